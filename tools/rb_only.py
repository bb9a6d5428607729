"""A few rebuild sequences of the bench shard (64 C2 live sets), for profilers: python tools/rb_only.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = _lib.Context(0)
sh = bench.Shard(ctx, bench.c2_problem(), 64, 2000, 45)
for i in range(reps):
    sh.rebuild()
ctx.sync()
