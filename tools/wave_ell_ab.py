#!/usr/bin/env python
"""k_ell_wave A/B: the batched MultiEllipsoid.update with DH_WAVE_ELL = 0 (off) / unset (default) / 1 / 2 on the
bench shard's live sets and on eggbox clouds: all outputs bit for bit, and the time per launch sequence."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
from dynesty_amd import _lib
ctx = _lib.Context(0)


def sets(name, runs):
    if name.startswith("mix"):  # three blobs in d dimensions, 3000 points
        d = int(name[3:])
        g = np.random.default_rng(d)
        c = g.uniform(0.25, 0.75, size=(3, d))
        pts = np.concatenate([c[i] + 0.02 * g.standard_normal((1000, d)) for i in range(3)])
    else:
        pts = inputs.cloud(name)
    n = pts.shape[0]
    return np.ascontiguousarray(np.stack([pts[np.random.default_rng(r).permutation(n)] for r in range(runs)]))


def run(allp, env, reps=10):
    os.environ.pop("DH_WAVE_ELL", None)
    if env is not None:
        os.environ["DH_WAVE_ELL"] = env
    r = ctx.rebuild_many(list(allp), multi=True)
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    dev = ctx.to_device(allp)
    runs, n, d = allp.shape
    me = max(1, n // (2 * d))
    bufs = [ctx.malloc(runs * 4), ctx.malloc(runs * 4), ctx.malloc(runs * me * d * 8), ctx.malloc(runs * me * d * d * 8),
            ctx.malloc(runs * me * d * d * 8), ctx.malloc(runs * me * d * d * 8), ctx.malloc(runs * me * d * 8),
            ctx.malloc(runs * me * 8)]
    go = lambda: ctx._check(ctx.lib.dh_rebuild_batch_dev(ctx.handle, runs, dev, n, d, 0, me, *bufs, None, None))
    go(); ctx.sync()
    ctx.record(e0)
    for _ in range(reps):
        go()
    ctx.record(e1)
    ms = ctx.elapsed_ms(e0, e1) / reps
    for b in bufs + [dev]:
        ctx.free(b)
    return r, ms


for name, runs in (("c2", 64), ("c3", 16), ("c3", 64), ("two5", 64), ("g3", 64), ("mix8", 64), ("mix10", 64), ("mix13", 64), ("ring2", 64)):
    allp = sets(name, runs)
    base, ms0 = run(allp, "0")
    line = [f"{name} x{runs} nells={base[0]['nells']}: off {ms0:.3f} ms"]
    for env in (None, "1", "2"):
        try:
            r, ms = run(allp, env)
        except Exception as ex:  # noqa: BLE001
            line.append(f"{env}: {type(ex).__name__} {ex}")
            continue
        same = all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for a, b in zip(r, base) for k in a)
        line.append(f"{'default' if env is None else env}: {ms:.3f} ms same={same}")
    print("  ".join(line), flush=True)
