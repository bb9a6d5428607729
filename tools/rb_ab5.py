"""Rebuild time of the bench shard (64 x 2000 x 25) and of 16 / 64 eggbox-like live sets under the current
environment: python tools/rb_ab5.py [reps].  One JSON line."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, inputs  # noqa: E402
from dynesty_amd import _lib  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = _lib.Context(0)
out = {}
s = bench.Shard(ctx, bench.c2_problem(), runs=64, seed=1000)
for i in range(10):
    s.rebuild()
ctx.sync()
ev = [ctx.event(), ctx.event()]
ctx.record(ev[0])
for i in range(reps):
    s.rebuild()
ctx.record(ev[1]); ctx.sync()
out["bench64_ms"] = round(ctx.elapsed_ms(ev[0], ev[1]) / reps, 4)
lib, h = ctx.lib, ctx.handle
def cloud_time(name, runs):
    pts = inputs.cloud(name); n, d = pts.shape
    allp = np.concatenate([pts[np.random.default_rng(r).permutation(n)] for r in range(runs)])
    d_p = ctx.to_device(allp); me = max(1, n // (2 * d))
    d_ne = ctx.malloc(runs * 4); d_st = ctx.malloc(runs * 4); d_nn = ctx.malloc(runs * 4)
    d_c = ctx.malloc(runs * me * d * 8); d_cov = ctx.malloc(runs * me * d * d * 8); d_am = ctx.malloc(runs * me * d * d * 8)
    d_ax = ctx.malloc(runs * me * d * d * 8); d_al = ctx.malloc(runs * me * d * 8); d_lv = ctx.malloc(runs * me * 8)
    def go():
        ctx._check(lib.dh_rebuild_batch_dev(h, runs, d_p, n, d, 0, me, d_ne, d_st, d_c, d_cov, d_am, d_ax, d_al, d_lv, None, d_nn))
    for _ in range(3):
        go()
    ctx.sync(); ctx.record(ev[0])
    for _ in range(10):
        go()
    ctx.record(ev[1]); ctx.sync()
    ne = ctx.from_device(d_ne, (runs,), np.int32); nn = ctx.from_device(d_nn, (runs,), np.int32)
    st = ctx.from_device(d_st, (runs,), np.int32)
    return dict(ms=round(ctx.elapsed_ms(ev[0], ev[1]) / 10, 4), nells=int(ne[0]), nodes=int(nn[0]), ok=bool((st == 0).all()))
for name, runs in (("c3", 16), ("c3", 64), ("two5", 64), ("c2", 64)):
    out[f"{name}x{runs}"] = cloud_time(name, runs)
print(json.dumps(out))
