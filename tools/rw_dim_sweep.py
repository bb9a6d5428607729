"""Wall time of one rwalk launch (32768 walkers x 200 steps, correlated Normal in a box) per dimension and kernel
form, through the host entry point (the copies are ~1 % of it): where does the four-lanes-per-walker form start to pay?
python tools/rw_dim_sweep.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems  # noqa: E402

ctx = _lib.Context(0)
k, walks = 32768, 200
for d in (2, 4, 5, 6, 8, 9, 12, 16):
    prob = problems.gauss_corr(d, 0.4, 5.0, f"c{d}")
    rng = np.random.default_rng(d)
    u0 = np.clip(0.5 + 0.03 * rng.standard_normal((k, d)), 0.01, 0.99)
    axes = 0.05 * np.linalg.qr(rng.standard_normal((d, d)))[0]
    st = ctx.seed_children([3, d], 0, k)
    row = {}
    for form in (1, 2):
        ctx.set_rwalk_form(form)
        ctx.rwalk_batch(prob, u0, axes, 1.0, -1e300, walks, st)
        ctx.rwalk_batch(prob, u0, axes, 1.0, -1e300, walks, st)
        t0 = time.perf_counter()
        for _ in range(5):
            out = ctx.rwalk_batch(prob, u0, axes, 1.0, -1e300, walks, st)
        row[form] = (time.perf_counter() - t0) / 5 * 1e3
    print(f"D={d:2d}  lane {row[1]:7.3f} ms   form-2 {row[2]:7.3f} ms   ({k * walks / row[1] / 1e6:.2f} / {k * walks / row[2] / 1e6:.2f} G proposals/s)")
