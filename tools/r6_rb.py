"""Rebuild time of the bench shard (R x 2000 x 25) the way bench.py brackets it -- HIP events around ONE rebuild + enlarge,
a sync between repetitions -- and back to back: python tools/r6_rb.py [reps] [R ...].  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Rs = [int(x) for x in sys.argv[2:]] or [1, 64, 128]
ctx = _lib.Context(0)
out = {}
ev = [ctx.event(), ctx.event()]
for R in Rs:
    s = bench.Shard(ctx, bench.c2_problem(), runs=R, seed=1000)
    for i in range(5):
        s.rebuild()
    ctx.sync()
    ts = []
    for i in range(reps):
        ctx.record(ev[0]); s.rebuild(); ctx.record(ev[1]); ctx.sync()
        ts.append(ctx.elapsed_ms(ev[0], ev[1]))
    ctx.record(ev[0])
    for i in range(reps):
        s.rebuild()
    ctx.record(ev[1]); ctx.sync()
    b = s.fetch_bound()
    out[f"runs{R}"] = dict(isolated_ms=round(float(np.median(ts)), 4), min_ms=round(float(np.min(ts)), 4),
                           back_to_back_ms=round(ctx.elapsed_ms(ev[0], ev[1]) / reps, 4),
                           ok=bool((b["status"] == 0).all()), nells=int(b["nells"][0]))
    del s
print(json.dumps(out))
