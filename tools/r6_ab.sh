# A/B of rebuild libraries on one box: bash tools/r6_ab.sh <tag> lib1.so lib2.so ...   (names inside dynesty_amd/)
O=gpurun_out/${1:-r6ab}; mkdir -p $O; shift
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib" | tee -a $O/ab.txt
  DYNHIP_LIB=$PWD/dynesty_amd/$lib timeout 300 python tools/r6_rb.py 30 1 64 128 2>&1 | grep runs | tee -a $O/ab.txt
done
done
