# full GPU suite + the default bench line (what the driver runs at round end)
O=gpurun_out/${1:-r5full}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
for k in ("roofline","roofline_rebuild","cpu_baseline"):
    print(k, {a:b for a,b in d[k].items() if not isinstance(b,(str,dict,list))})
c=d["config"]
for k in c:
    if k in ("throughput_rng_mode","end_to_end","verified") or "c3" in k.lower() or "c4" in k.lower():
        print(k, json.dumps(c[k])[:1500])
PY
