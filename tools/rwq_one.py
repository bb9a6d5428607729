"""A few launches of the rwalk kernel at the bench's launch shape (64 x 512 walkers x 45 steps), for profilers:
python tools/rwq_one.py [form] [reps]   (form 2 = four lanes per walker, 1 = one walker per lane)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

form = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _lib.Context(0)
ctx.set_rwalk_form(form)
prob = bench.c2_problem()
sh = bench.Shard(ctx, prob, 64, 2000, 45)
sh.rebuild()
kq = 512
idxq = (np.arange(64 * kq, dtype=np.int32) // kq) * bench.MAX_ELLS
ctx._check(ctx.lib.dh_memcpy_h2d(ctx.handle, sh.d_idx, idxq.ctypes.data, idxq.nbytes))
for i in range(reps):
    sh.walk(i, 0, 64 * kq)
ctx.sync()
