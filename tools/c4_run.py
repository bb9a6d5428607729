#!/usr/bin/env python
"""BASELINE config C4 on the device: 200-D iid Normal, Normal prior (ndtri),
bound='single', sample='rslice' (slices = 203), nlive = 4000.
Reference: 74.3 k calls/s, 68.9 it/s, not converged in 400 s."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from dynesty_amd import nested, problems, _lib
maxiter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
prob = problems.gauss_normal_prior(200, "C4")
ctx = _lib.default_context(0)
pts = 0.5 + 0.05 * np.random.default_rng(1).standard_normal((4000, 200))
t = time.perf_counter(); ctx.rebuild(pts, multi=False); t1 = time.perf_counter() - t
t = time.perf_counter(); ctx.rebuild(pts, multi=False); t2 = time.perf_counter() - t
print(json.dumps(dict(rebuild_4000x200_first_s=t1, rebuild_4000x200_s=t2)))
t = time.perf_counter()
r = nested.run_static(prob, nlive=4000, bound='single', sample='rslice', slices=203,
                      queue_size=K, rstate=np.random.default_rng(21), dlogz=0.01,
                      maxiter=maxiter if maxiter > 0 else None)
dt = time.perf_counter() - t
print(json.dumps(dict(niter=r.niter, ncall=r.ncall, seconds=dt, calls_per_s=r.ncall / dt,
                      it_per_s=r.niter / dt, logz=r.logz, logzerr=r.logzerr, nbound=r.nbound,
                      truth=prob.logz_truth, converged=(maxiter <= 0 or r.niter < maxiter))))
