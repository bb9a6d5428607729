#!/usr/bin/env python
"""Reference runs behind tests/golden/c2_logz_ref.json: the REAL dynesty (from /root/reference/py, build
container only) on BASELINE config C2 -- 25-D rho=0.4 correlated Normal, nlive=2000, bound='multi',
sample='rwalk' (walks defaults to 20 + ndim = 45), dlogz=0.01 -- at queue size K.

  K = 1: the plain serial sampler (sampler.py:696-699, one shared generator)
  K > 1: `SerialPool(K)`: the reference's exact K-in-flight semantics
         (sampler.py:690-778), executed serially.

usage: ref_c2_runs.py K seed [seed ...]   -> one JSON line per seed on stdout
"""
import json
import sys
import time

sys.path.insert(0, "/root/repo/tests")
sys.path.insert(0, "/root/repo")
import refshim  # noqa: E402

refshim.import_reference()
import numpy as np  # noqa: E402
import dynesty  # noqa: E402
from dynesty_amd import problems  # noqa: E402  (host callables only: no device needed)


class SerialPool:
    def __init__(self, size):
        self.size = size

    def map(self, f, x):
        return list(map(f, x))


K = int(sys.argv[1])
prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
for seed in map(int, sys.argv[2:]):
    t = time.time()
    kw = {}
    if K > 1:
        kw = dict(pool=SerialPool(K), queue_size=K)
    s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 25, nlive=2000, bound='multi',
                              sample='rwalk', rstate=np.random.default_rng(seed), **kw)
    s.run_nested(dlogz=0.01, print_progress=False)
    r = s.results
    print(json.dumps(dict(K=K, seed=seed, logz=float(r.logz[-1]), logzerr=float(r.logzerr[-1]),
                          niter=int(r.niter), ncall=int(np.sum(r.ncall)), truth=prob.logz_truth,
                          seconds=time.time() - t)), flush=True)
