"""Cycle split of rwalkq_kernel at the bench's launch shape (64 x 512 walkers x 45 steps): generator fill
against the rest of a step, rounds / scalar segments / wedge tests per step.  Needs the instrumented build:
make -C dynesty_amd/csrc wqprof;  DYNHIP_LIB=dynesty_amd/libdynhip_wqprof.so python tools/wq_prof.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dynesty_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
ctx.set_rwalk_form(2)
prob = bench.c2_problem()
sh = bench.Shard(ctx, prob, 64, 2000, 45)
sh.rebuild()
kq = 512
idxq = (np.arange(64 * kq, dtype=np.int32) // kq) * bench.MAX_ELLS
ctx._check(ctx.lib.dh_memcpy_h2d(ctx.handle, sh.d_idx, idxq.ctypes.data, idxq.nbytes))
for i in range(5):
    sh.walk(i, 0, 64 * kq)
ctx.sync()
wk = sh.fetch_walk()
v = wk["v"][:64 * kq].reshape(-1, 16, 25)[:, 0, :9]  # lane 0 of every wavefront
steps = 45
print(json.dumps(dict(fill_cycles_per_step=float(v[:, 0].mean() / steps), rest_cycles_per_step=float(v[:, 1].mean() / (steps - 1)),
                      rounds_per_step=float(v[:, 2].mean() / steps), segs_per_round=float((v[:, 3] / v[:, 2]).mean()),
                      wedges_per_round=float((v[:, 4] / v[:, 2]).mean())), indent=1))
