#!/usr/bin/env python
"""Stand-alone durations of the rebuild and the walk of a sub-shard of 64/G runs (one stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
prob = bench.c2_problem()
ctx = _lib.Context(0)
for g in (1, 2, 4, 8):
    s = bench.Shard(ctx, prob, runs=64 // g, seed=1000)
    for i in range(60 if g == 1 else 10):
        s.step(i)
    ctx.sync()
    out = []
    for what in ("rebuild", "walk"):
        t = time.perf_counter()
        for i in range(30):
            s.rebuild() if what == "rebuild" else s.walk(i)
        ctx.sync()
        out.append(1e3 * (time.perf_counter() - t) / 30)
    print(f"runs={64 // g}: rebuild {out[0]:.3f} ms  walk {out[1]:.3f} ms", flush=True)
