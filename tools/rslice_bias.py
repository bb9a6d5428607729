#!/usr/bin/env python
"""logZ of the iid-Normal / Normal-prior problem (the C4 family) at dimension D with
bound='single', sample='rslice' on the device, for comparison with the reference run on the CPU
with the same settings (the offset from the analytic logZ is a property
of the sampler settings, not of the device path)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from dynesty_amd import nested, problems

D = int(sys.argv[1]); nlive = int(sys.argv[2]); K = int(sys.argv[3])
seeds = [int(s) for s in sys.argv[4:]] or [1]
prob = problems.gauss_normal_prior(D, "C4")
for seed in seeds:
    t = time.perf_counter()
    r = nested.run_static(prob, nlive=nlive, bound='single', sample='rslice', queue_size=K,
                          rstate=np.random.default_rng(seed), dlogz=0.01)
    print(json.dumps(dict(D=D, nlive=nlive, K=K, seed=seed, logz=r.logz, logzerr=r.logzerr, niter=r.niter,
                          ncall=r.ncall, truth=prob.logz_truth, seconds=time.perf_counter() - t)))
