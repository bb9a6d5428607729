# rebuild A/B of the current tree: timing + the rebuild parity tests.  bash tools/r5_rb2.sh <tag> [notest]
O=gpurun_out/${1:-r5rb2}; mkdir -p $O
timeout 300 python tools/rb_ab5.py 60 2>&1 | tee $O/rb_ab.txt
timeout 300 python tools/rb_ab5.py 60 2>&1 | tee -a $O/rb_ab.txt
if [ -z "$2" ]; then
timeout 900 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py tests/test_gpu_livesets.py tests/test_gpu_bench_shape.py tests/test_gpu_small_kernels.py -x -q 2>&1 | tail -3 | tee -a $O/rb_ab.txt
fi
