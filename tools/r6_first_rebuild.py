import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
i0 = [i for i, r in enumerate(rows) if "k_root_parts" in r["Kernel_Name"]][0]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 6):i0 + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us +%9.1f us %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]))
