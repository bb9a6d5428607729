"""Print the top rows of a rocprofv3 kernel_stats CSV: python tools/kstats.py FILE [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "")[:64]
    print("%-64s calls=%6s avg_us=%9.1f tot_ms=%9.2f %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
