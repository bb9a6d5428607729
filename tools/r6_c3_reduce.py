"""Timeline of one mid-run rebuild of a traced resident loop + per-kernel totals between two rebuilds: python tools/r6_c3_reduce.py DIR"""
import csv, glob, sys
from collections import defaultdict
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
roots = [i for i, r in enumerate(rows) if "k_root_parts" in r["Kernel_Name"]]
print("rebuilds", len(roots))
a, b = roots[len(roots) * 3 // 4], roots[len(roots) * 3 // 4 + 1]
t0 = int(rows[a]["Start_Timestamp"])
tot = defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n:32s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
print("period between the two rebuilds: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:36s} {c:4d} x  {us / c:8.1f} us = {us:9.1f} us")
