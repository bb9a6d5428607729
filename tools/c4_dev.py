#!/usr/bin/env python
"""BASELINE config C4 through the device-resident loop (dh_ns_ensemble above the register-resident dimensions):
200-D iid Normal, Normal prior, bound='single', sample='rslice' (slices = 203), nlive = 4000.
usage: c4_dev.py [K] [runs]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import problems, _lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prob = problems.gauss_normal_prior(200, "C4")
ctx = _lib.default_context(0)
ctx.ns_ensemble(prob, 1, 4000, K, bound='single', sample='rslice', slices=203, entropy=[3], max_fills=3, max_iter=250000)
for rep in range(2):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 4000, K, bound='single', sample='rslice', slices=203, entropy=[21 + rep],
                        max_iter=250000)
    dt = time.perf_counter() - t
    print(json.dumps(dict(K=K, runs=runs, seconds=round(dt, 3), logz=r["logz"].tolist(), logzerr=r["logzerr"].tolist(),
                          niter=r["niter"].tolist(), ncall=r["ncall"].tolist(), nbound=r["nbound"].tolist(),
                          status=r["status"].tolist(), nfills=r["nfills"], truth=prob.logz_truth,
                          calls_per_s=float(r["ncall"].sum() / dt))))
