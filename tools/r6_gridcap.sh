# capped level grids (a workgroup loops over its run's parts / children) against the worst-case grids
cd /root/repo; O=gpurun_out/${1:-r6gridcap}; mkdir -p $O
for rep in 1 2; do
for mode in 1 0; do
  echo "== DH_LEVEL_GRID_CAP=$mode" | tee -a $O/ab.txt
  DH_LEVEL_GRID_CAP=$mode timeout 300 python tools/r6_rb.py 30 1 16 64 128 2>&1 | grep runs | tee -a $O/ab.txt
  DH_LEVEL_GRID_CAP=$mode timeout 300 python tools/ns_c3.py 16 2>&1 | tail -1 | tee -a $O/ab.txt
done
done
timeout 1500 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py tests/test_gpu_bench_shape.py tests/test_gpu_livesets.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/ab.txt
