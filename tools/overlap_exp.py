#!/usr/bin/env python
"""Experiment: the bench step (64-run rebuild + 128 k rwalk walkers) as G sub-shards on G contexts
(each its own HIP stream), issued in phase (every sub-shard: rebuild then walk) or phase-shifted
(odd sub-shards: walk on the previous step's bound, then rebuild).  usage: overlap_exp.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prob = bench.c2_problem()


def run(groups, shifted):
    ctxs = [_lib.Context(0) for _ in range(groups)]
    shards = [bench.Shard(c, prob, runs=64 // groups, seed=1000 + g) for g, c in enumerate(ctxs)]
    for i in range(60):  # clock ramp + warm-up
        for s in shards:
            s.step(i)
    for c in ctxs:
        c.sync()
    t = time.perf_counter()
    for i in range(steps):
        for g, s in enumerate(shards):
            if shifted and g % 2 == 1:
                s.walk(i)
                s.rebuild()
            else:
                s.rebuild()
                s.walk(i)
    t_issue = time.perf_counter() - t
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t
    ok = all(int(s.fetch_bound()["status"].min()) == 0 for s in shards)
    print(f"groups={groups} shifted={int(shifted)}  ms/step={1e3 * dt / steps:.3f}  host issue ms/step="
          f"{1e3 * t_issue / steps:.3f}  status_ok={ok}", flush=True)


CASES = ((1, False), (2, False), (2, True), (4, False), (4, True), (8, True), (1, False))
if len(sys.argv) > 3:
    CASES = ((int(sys.argv[2]), bool(int(sys.argv[3]))),)
for groups, shifted in CASES:
    run(groups, shifted)
