"""The bench's walk launch (64 runs x 512 walkers x 45 steps, 25-D) in both RNG modes, for rocprofv3 --kernel-trace
--stats: per-kernel time of the generator pass and of the walk.  python tools/rng_launch_prof.py [reps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = _lib.Context(0)
sh = bench.Shard(ctx, bench.c2_problem(), 64, 2000, 45)
sh.rebuild()
ev = [ctx.event() for _ in range(2)]
out = {}
for name, fn in (("pcg64", lambda i: sh.walk_q(i, 0)), ("philox", lambda i: sh.walk_q_philox(i, 0))):
    for i in range(20):
        fn(i)
    ctx.sync()
    ctx.record(ev[0])
    for i in range(reps):
        fn(i)
    ctx.record(ev[1])
    ctx.sync()
    out[name + "_ms"] = round(ctx.elapsed_ms(ev[0], ev[1]) / reps, 4)
print(json.dumps(out))
