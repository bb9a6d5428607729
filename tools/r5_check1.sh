# round 5, first GPU check: generator passes (itemgen at 64 VGPRs, Philox item pass) -- parity tests, kernel times, bench line
set -x
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rwalkq.py tests/test_gpu_philox.py tests/test_gpu_rwalk.py tests/test_gpu_rng.py tests/test_gpu_bench_shape.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/rwq_time.py 30 > $O/rwq_time.json 2>&1
cat $O/rwq_time.json
timeout 600 python bench.py --lean --steps 100 --warmup 5 > $O/bench_lean.json 2> $O/bench_lean.err
cat $O/bench_lean.json | head -c 3000
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o bench -- python $GRAFT_REPO_ROOT/bench.py --lean --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.log
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python tools/kstats.py $(find $O -name "*kernel_stats.csv" | head -1) 2>/dev/null | head -30
