# the root of more runs than fit at once: chunks of co-resident runs / one launch / the single-workgroup fallback
cd /root/repo; O=gpurun_out/${1:-r6rootchunk}; mkdir -p $O
for rep in 1 2; do
for mode in "DH_X=0" "DH_ROOT_ONE_LAUNCH=1" "DH_ROOT_CHUNK=0"; do
  echo "== $mode" | tee -a $O/ab.txt
  env $mode timeout 300 python tools/r6_rb.py 30 64 128 160 256 2>&1 | grep runs | tee -a $O/ab.txt
done
done
timeout 1200 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/ab.txt
