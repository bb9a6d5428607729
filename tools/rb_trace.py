"""Summarise the kernel trace of one batched rebuild: python tools/rb_trace.py <kernel_trace.csv> [which]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
roots = [i for i, r in enumerate(rows) if 'k_root' in r['Kernel_Name']]
a = roots[which]
b = next(i for i in range(a, len(rows)) if 'k_finish' in rows[i]['Kernel_Name'])
t0 = int(rows[a]['Start_Timestamp'])
idle = 0
for r in rows[a:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    s = int(r['Start_Timestamp']) - t0
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if d > 12000:
        print(f"{s/1e3:9.1f} us +{d/1e3:8.1f}  {n}  wgs={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}")
    else:
        idle += d
print(f"idle-level launches: {idle/1e3:.1f} us ; total {(int(rows[b]['End_Timestamp'])-t0)/1e3:.1f} us")
