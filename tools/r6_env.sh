# environment experiments on the current library: bash tools/r6_env.sh <tag> "VAR=val ..." "VAR=val ..." ...
O=gpurun_out/${1:-r6env}; mkdir -p $O; shift
for e in "$@"; do
  echo "== $e" | tee -a $O/env.txt
  env $e timeout 300 python tools/r6_rb.py 20 1 64 128 2>&1 | grep runs | tee -a $O/env.txt
done
