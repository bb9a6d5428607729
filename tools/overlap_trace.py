"""Concurrency seen in a kernel trace: per queue busy time, union busy time, pairwise overlap.
python tools/overlap_trace.py <kernel_trace.csv> [last_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * (1 - frac)):]
t0 = int(rows[0]['Start_Timestamp'])
t1 = max(int(r['End_Timestamp']) for r in rows)
byq = {}
for r in rows:
    byq.setdefault(r.get('Queue_Id', '?'), []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
print(f"window {(t1 - t0) / 1e6:.3f} ms, {len(rows)} kernels, queues: { {q: len(v) for q, v in byq.items()} }")
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1))
    ev.append((int(r['End_Timestamp']), -1))
ev.sort()
depth = 0; last = t0; hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
tot = sum(hist.values())
print("time by number of kernels in flight:", {k: f"{100 * v / tot:.1f}%" for k, v in sorted(hist.items())})
# rwalk kernels: who runs beside them
for r in rows[-400:]:
    if 'rwalk' in r['Kernel_Name']:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        others = [(o['Kernel_Name'].split('::')[-1].split('(')[0][:14], (min(e, int(o['End_Timestamp'])) - max(s, int(o['Start_Timestamp']))) / 1e3)
                  for o in rows if o is not r and int(o['Start_Timestamp']) < e and int(o['End_Timestamp']) > s]
        print(f"rwalk q={r.get('Queue_Id')} {(e - s) / 1e3:.0f} us; beside it: {others[:12]}")
