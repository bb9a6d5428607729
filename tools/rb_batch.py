"""Batched rebuild of R permuted copies of a cloud: python tools/rb_batch.py <cloud> <runs> [reps] [mode]"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
from dynesty_amd import _lib
ctx = _lib.Context(0); lib = ctx.lib; h = ctx.handle
name = sys.argv[1]; runs = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pts = inputs.cloud(name); n, d = pts.shape
allp = np.concatenate([pts[np.random.default_rng(r).permutation(n)] for r in range(runs)])
d_p = ctx.to_device(allp); me = max(1, n // (2 * d))
d_ne = ctx.malloc(runs * 4); d_st = ctx.malloc(runs * 4); d_nn = ctx.malloc(runs * 4)
d_c = ctx.malloc(runs * me * d * 8); d_cov = ctx.malloc(runs * me * d * d * 8); d_am = ctx.malloc(runs * me * d * d * 8)
d_ax = ctx.malloc(runs * me * d * d * 8); d_al = ctx.malloc(runs * me * d * 8); d_lv = ctx.malloc(runs * me * 8)
e0, e1 = ctx.event(), ctx.event()
for r in range(reps + 1):
    if r == 1: ctx.record(e0)
    ctx._check(lib.dh_rebuild_batch_dev(h, runs, d_p, n, d, mode, me, d_ne, d_st, d_c, d_cov, d_am, d_ax, d_al, d_lv, None, d_nn))
ctx.record(e1); ms = ctx.elapsed_ms(e0, e1) / reps
st = ctx.from_device(d_st, (runs,), np.int32); ne = ctx.from_device(d_ne, (runs,), np.int32)
print(f"{name} runs={runs} mode={mode}: {ms:.3f} ms/launch  nells={ne[:4]} bad={int((st != 0).sum())}")
