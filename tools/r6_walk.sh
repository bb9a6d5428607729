# walk kernels after a change: parity tests, then the bench line (lean) twice
cd /root/repo; O=gpurun_out/${1:-r6walk}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rwalkq.py tests/test_gpu_philox.py tests/test_gpu_bench_shape.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --lean --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','ms_per_step')}, 'walk launch ms', r['roofline']['kernel_ms'])" | tee -a $O/bench.txt
done
