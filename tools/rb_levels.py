#!/usr/bin/env python
"""Bench-shard rebuild of R runs, a few repetitions (for a rocprofv3 --kernel-trace + tools/rb_trace.py timeline)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = _lib.Context(0)
s = bench.Shard(ctx, bench.c2_problem(), runs=R, seed=1000)
for i in range(30):
    s.rebuild()
ctx.sync()
