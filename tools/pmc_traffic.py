"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, each with
--kernel-trace only, as MI355X_MICROARCH.md prescribes):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python bench.py --lean --preroll 0 --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -- python bench.py --lean --preroll 0 --steps 3 --warmup 1
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/r03/pmc_traffic.json

gfx950 correction: FETCH_SIZE counts 128-byte requests as 64 B -> x2; both counters are in KB."""
import csv, glob, json, sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[name].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --lean "
                  "--preroll 0 --steps 3 --warmup 1 (two separate passes)",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section); "
                     "WRITE_SIZE uncorrected; both in KB",
       "kernels": {}}
# the walk kernel of the bench's launch shape: four lanes per walker at the gate queue (round 3), else one per lane
rw = [k for k in fetch if k.startswith("rwalkq_kernel<7")] or [k for k in fetch if k.startswith("rwalk_kernel<25, true, 1")]
# round 4: a walk launch = the generator pass + the walk kernel
ig = [k for k in fetch if k.startswith("itemgen_kernel")]
steps = len(fetch.get(rw[0], [])) if rw else 1
for name in sorted(set(fetch) | set(write)):
    fk = sum(v for _, v in fetch.get(name, []))
    wk = sum(v for _, v in write.get(name, []))
    n = max(len(fetch.get(name, [])), len(write.get(name, [])))
    out["kernels"][name] = {"launches": n, "FETCH_SIZE_KB_per_launch_raw": fk / max(n, 1),
                            "WRITE_SIZE_KB_per_launch_raw": wk / max(n, 1),
                            "traffic_bytes_per_launch": (2 * fk + wk) * 1024 / max(n, 1),
                            "traffic_bytes_per_bench_step": (2 * fk + wk) * 1024 / steps}
rb = [k for k in out["kernels"] if k.split("<")[0] in ("k_root_parts", "k_split", "k_ell", "k_finish", "k_out_eig",
                                                      "k_root_eig", "k_tree")]
out["rwalk_launches_profiled"] = steps
if rw:
    out["rwalk_launch_traffic_bytes"] = sum(out["kernels"][k]["traffic_bytes_per_launch"] for k in rw[:1] + ig[:1])
    out["rwalk_launch_kernels"] = rw[:1] + ig[:1]
nrb = out["kernels"].get("k_root_parts", {}).get("launches", 0)
out["rebuild_pipelines_profiled"] = nrb
# one pipeline = k_root_parts (+ k_root_eig on the side stream) + levels x (k_split, k_ell<false>) + k_tree + k_finish + k_out_eig
out["rebuild_pipeline_bytes_per_launch_sequence"] = sum(
    out["kernels"][k]["traffic_bytes_per_launch"] * out["kernels"][k]["launches"] for k in rb) / max(nrb, 1)
for k in out["kernels"].values():
    del k["traffic_bytes_per_bench_step"]
print(json.dumps(out, indent=1))
