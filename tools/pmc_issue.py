"""Issue / stall counters per kernel from rocprofv3 --pmc passes (SQ block, 8 counters a pass, --kernel-trace only):

    python tools/pmc_issue.py gpurun_out/<dir> > profiles/r04/pmc_issue.json      (<dir> holds <workload>_<pass>/ subdirs,
                                                                                   see tools/prof_r04.sh)

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts;
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (stalled at issue: dependency, pipe busy) + ACTIVE_INST_ANY
~ WAVE_CYCLES.  valu_active_frac = ACTIVE_INST_VALU / WAVE_CYCLES: the share of a wavefront's life in which it is
issuing a vector instruction."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
# `reduce <dir>`: fold a pass's raw counter_collection.csv (one row per dispatch and counter: tens of MB for a resident
# loop) into per-kernel means next to it and delete the raw files (gpurun copies at most 64 MiB back)
if root == "reduce":
    d = sys.argv[2]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            e = acc[r["Kernel_Name"]][r["Counter_Name"]]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    with open(os.path.join(d, "reduced.json"), "w") as fo:
        json.dump({k: {c: [v[0] / v[1], v[1]] for c, v in cs.items()} for k, cs in acc.items()}, fo)
    for f in glob.glob(d + "/*/*.csv") + glob.glob(d + "/*.csv") + glob.glob(d + "/*/*.db"):
        os.remove(f)
    sys.exit(0)
KEEP = ("rwalkq_kernel", "itemgen_kernel", "rwalk_kernel", "k_ell<false", "k_split", "k_root_parts", "k_root_eig", "k_finish",
        "wide_walk_kernel", "wide_eig2_kernel", "ns_consume", "ns_select", "cube_quad_kernel", "contains_runs_kernel",
        "slice_kernel", "unif_kernel")


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


out = {"units": "quad-cycles summed over wavefronts (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*); instruction counts "
                "summed over wavefronts; per launch averages",
       "absent_on_gfx950_rocprofv3": ["SQ_INSTS_VALU_MFMA_F64 (present as SQ_INSTS_VALU_MFMA_MOPS_F64 / SQ_INSTS_MFMA)"],
       "workloads": {}}
for wl in sorted({os.path.basename(d).rsplit("_", 1)[0] for d in glob.glob(os.path.join(root, "*_p[0-9]"))}):
    per = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, wl + "_p[0-9]"))):
        red = json.load(open(os.path.join(d, "reduced.json")))
        for name, cs in red.items():
            k = short(name)
            if not k.startswith(KEEP):
                continue
            for c, (mean, n) in cs.items():
                per[k][c] = [mean] * int(n)
    res = {}
    for k, cs in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {"launches_profiled": max(len(v) for v in cs.values()), "counters_per_launch": {c: round(x, 1) for c, x in sorted(m.items())}}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            e["valu_active_frac"] = round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
            e["issue_active_frac"] = round(m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
            e["wait_waitcnt_frac"] = round(m.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            e["wait_issue_stall_frac"] = round(m.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            if "SQ_WAIT_INST_LDS" in m:
                e["wait_issue_lds_frac"] = round(m["SQ_WAIT_INST_LDS"] / wc, 4)
            if "SQ_WAVES" in m and m["SQ_WAVES"]:
                e["cycles_per_wavefront"] = round(4 * wc / m["SQ_WAVES"], 1)
        ins = sum(m.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        if ins and wc:
            e["cycles_per_instruction_per_wavefront"] = round(4 * wc / ins, 2)
        res[k] = e
    out["workloads"][wl] = res
print(json.dumps(out, indent=1))
