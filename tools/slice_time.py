"""Kernel time of the fused rslice / slice kernels at D = 25 (C2 problem, 64 frames): python tools/slice_time.py"""
import os, sys, time, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from dynesty_amd import _lib
ctx = _lib.Context(0)
case = inputs.walker_case("C2", 40000, 7)
prob = case["problem"]
k = 32768
u0 = case["u0"][np.arange(k) % len(case["u0"])]
axes = np.stack([case["axes"]] * 64)
idx = (np.arange(k) // 512).astype(np.int32)
states = ctx.seed_children([5, 5], 0, k)
for name, kw in (("rslice", dict(principal=False)), ("slice", dict(principal=True))):
    ctx.slice_batch(prob, u0, axes, 0.5, case["loglstar"], 5 if name == "rslice" else 1, states, axes_idx=idx, **kw)
    t = time.perf_counter()
    for _ in range(3):
        r = ctx.slice_batch(prob, u0, axes, 0.5, case["loglstar"], 5 if name == "rslice" else 1, states, axes_idx=idx, **kw)
    dt = (time.perf_counter() - t) / 3
    print(json.dumps(dict(kernel=name, lib=os.environ.get("DYNHIP_LIB", "default"), walkers=k, ms_incl_transfers=round(dt * 1e3, 3),
                          ncalls=int(r["ncalls"].sum()))))
