#!/usr/bin/env python
"""Reference runs behind tests/golden/c3_logz_ref.json: the REAL dynesty (from /root/reference/py, build
container only) on BASELINE config C3 -- eggbox 2-D, nlive = 5000, bound='multi', sample='rslice'
(slices = 3 + ndim = 5), dlogz = 0.01 -- at queue size K.

  K = 1: the plain serial sampler (sampler.py:696-699, one shared generator)
  K > 1: `SerialPool(K)`: the reference's exact K-in-flight semantics
         (sampler.py:690-778), executed serially.

usage: ref_c3_runs.py K seed [seed ...]   -> one JSON line per seed on stdout (about 65 s per run)"""
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
prob = problems.eggbox(2, name="C3")
for seed in map(int, sys.argv[2:]):
    t = time.time()
    kw = dict(pool=SerialPool(K), queue_size=K) if K > 1 else {}
    s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 2, nlive=5000, bound='multi',
                              sample='rslice', rstate=np.random.default_rng(seed), **kw)
    s.run_nested(dlogz=0.01, print_progress=False)
    r = s.results
    print(json.dumps(dict(K=K, seed=seed, logz=float(r.logz[-1]), logzerr=float(r.logzerr[-1]),
                          niter=int(r.niter), ncall=int(np.sum(r.ncall)), nbound=int(len(r.bound)) if hasattr(r, 'bound') else -1,
                          seconds=time.time() - t)), flush=True)
