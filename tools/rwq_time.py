"""Kernel time of the two rwalk forms (one walker per lane / four lanes per walker) at the bench shard's
launch shapes: 64 x 2000 walkers (one bound-update interval in one launch) and 64 x 512 (the queue size
of the evidence gate).  python tools/rwq_time.py [reps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0)
prob = bench.c2_problem()
sh = bench.Shard(ctx, prob, 64, 2000, 45)
sh.rebuild()
ev = [ctx.event() for _ in range(2)]
kq = 512
idxq = (np.arange(64 * kq, dtype=np.int32) // kq) * bench.MAX_ELLS
out = {}
for form in (1, 2):
    ctx.set_rwalk_form(form)
    for name, count, idx in (("64x2000", None, sh.idx), ("64x512", 64 * kq, idxq)):
        ctx._check(ctx.lib.dh_memcpy_h2d(ctx.handle, sh.d_idx, idx.ctypes.data, idx.nbytes))
        for rng in ("pcg64", "philox"):
            if rng == "philox" and count is not None:
                continue
            fn = (lambda i: sh.walk(i, 0, count)) if rng == "pcg64" else (lambda i: sh.walk_philox(i))
            for i in range(30):
                fn(i)
            ctx.sync()
            ctx.record(ev[0])
            for i in range(reps):
                fn(i)
            ctx.record(ev[1])
            ctx.sync()
            ms = ctx.elapsed_ms(ev[0], ev[1]) / reps
            n = (sh.k if count is None else count) * 45
            wk = sh.fetch_walk()
            out[f"form{form}_{name}_{rng}"] = dict(ms=round(ms, 4), gprops_per_s=round(n / ms / 1e6, 3),
                                                   accept_frac=float(wk["accept"][:(sh.k if count is None else count)].mean() / 45))
print(json.dumps(out, indent=1))
