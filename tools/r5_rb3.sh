O=gpurun_out/${1:-r5rb3}; mkdir -p $O
echo "== default"; timeout 300 python tools/rb_ab5.py 60 2>&1 | tee $O/rb_ab.txt
for v in variants/*.so; do echo "== $v"; DYNHIP_LIB=$PWD/$v timeout 300 python tools/rb_ab5.py 60 2>&1 | tee -a $O/rb_ab.txt; done
timeout 900 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py tests/test_gpu_livesets.py tests/test_gpu_bench_shape.py tests/test_gpu_small_kernels.py -x -q 2>&1 | tail -3 | tee -a $O/rb_ab.txt
