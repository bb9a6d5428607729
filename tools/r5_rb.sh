O=gpurun_out/${1:-r5rb}; mkdir -p $O
for v in "DH_LEAF_SIDE=0" "DH_LEAF_SIDE=1" "DH_LEAF_FROM_PTS=100" "DH_LEAF_FROM_PTS=200"; do echo "== $v"; env $v timeout 300 python tools/rb_ab5.py 40; done 2>&1 | tee $O/rb_ab.txt
timeout 900 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_edges.py tests/test_gpu_livesets.py tests/test_gpu_bench_shape.py -x -q 2>&1 | tail -3 | tee -a $O/rb_ab.txt
