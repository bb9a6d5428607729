#!/usr/bin/env python
"""Experiment: the 64-run rebuild as G half-batches on G contexts (each its own stream and scratch), issued back to back so
that their level kernels overlap (a batch at its oversubscribed deep levels beside one at its underfilled top
levels).  usage: rebuild_split_exp.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prob = bench.c2_problem()


def run(groups):
    ctxs = [_lib.Context(0) for _ in range(groups)]
    shards = [bench.Shard(c, prob, runs=64 // groups, seed=1000 + g) for g, c in enumerate(ctxs)]
    for i in range(40):
        for s in shards:
            s.rebuild()
    for c in ctxs:
        c.sync()
    t = time.perf_counter()
    for i in range(steps):
        for s in shards:
            s.rebuild()
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t
    ok = all(int(s.fetch_bound()["status"].min()) == 0 for s in shards)
    print(f"groups={groups}  rebuild of 64 runs: {1e3 * dt / steps:.3f} ms  status_ok={ok}", flush=True)


for g in (1, 2, 4, 1, 2):
    run(g)
