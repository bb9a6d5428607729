#!/usr/bin/env python
"""Timeline (HIP events, no profiler) of G sub-shards on G streams, phase-shifted. usage: overlap_events.py G [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
prob = bench.c2_problem()
ctxs = [_lib.Context(0) for _ in range(G)]
shards = [bench.Shard(c, prob, runs=64 // G, seed=1000 + g) for g, c in enumerate(ctxs)]
for i in range(40):
    for s in shards:
        s.step(i)
for c in ctxs:
    c.sync()
evs = []
base = ctxs[0].event()
ctxs[0].record(base)
th = []
t0 = time.perf_counter()
for i in range(steps):
    for g, (c, s) in enumerate(zip(ctxs, shards)):
        seq = ("W", "R") if g % 2 == 1 else ("R", "W")
        for what in seq:
            s.rebuild() if what == "R" else s.walk(i)
            e = c.event(); c.record(e)
            evs.append((g, i, what, e, 1e3 * (time.perf_counter() - t0)))
for c in ctxs:
    c.sync()
rows = [(ctxs[0].elapsed_ms(base, e), g, i, what, th) for g, i, what, e, th in evs]
rows.sort()
last = {}
for t, g, i, what, th in rows:
    print(f"{t:8.3f} ms  group {g} step {i} {what} done  (+{t - last.get(g, 0.0):6.3f})   host issued at {th:7.3f}")
    last[g] = t
