# rebuild A/B of the current tree: timing (+ the rebuild parity tests unless a 2nd argument is given).  bash tools/r6_rb.sh <tag> [notest]
O=gpurun_out/${1:-r6rb}; mkdir -p $O
timeout 300 python tools/r6_rb.py 30 1 64 128 2>&1 | tee $O/rb.txt
if [ -z "$2" ]; then
timeout 1200 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py tests/test_gpu_livesets.py tests/test_gpu_bench_shape.py tests/test_gpu_small_kernels.py -x -q 2>&1 | tail -5 | tee -a $O/rb.txt
fi
