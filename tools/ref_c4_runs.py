#!/usr/bin/env python
"""Reference runs behind tests/golden/c4_logz_ref.json: the REAL dynesty (from /root/reference/py, build
container only) on BASELINE config C4 -- 200-D iid Normal likelihood, Normal prior through ndtri, nlive = 4000,
bound='single', sample='rslice' (slices = 3 + ndim = 203), dlogz = 0.01 -- at queue size K (K = 1: the serial
sampler; K > 1: pool = SerialPool(K)).  One run takes 45-60 minutes on one core.

usage: ref_c4_runs.py K seed [seed ...]   -> one JSON line per seed on stdout"""
import json
import sys
import time

sys.path.insert(0, "/root/repo/tests")
import refshim  # noqa: E402

refshim.import_reference()
import numpy as np  # noqa: E402
from scipy.special import ndtri  # noqa: E402
import dynesty  # noqa: E402

D, nlive = 200, 4000
c = -0.5 * D * np.log(2 * np.pi)


def loglike(x):
    return c - 0.5 * np.dot(x, x)


def ptform(u):
    return ndtri(u)


class SerialPool:
    def __init__(self, size):
        self.size = size

    def map(self, f, x):
        return list(map(f, x))


K = int(sys.argv[1])
for seed in map(int, sys.argv[2:]):
    t = time.time()
    kw = dict(pool=SerialPool(K), queue_size=K) if K > 1 else {}
    s = dynesty.NestedSampler(loglike, ptform, D, nlive=nlive, bound='single', sample='rslice',
                              rstate=np.random.default_rng(seed), **kw)
    s.run_nested(dlogz=0.01, print_progress=False)
    r = s.results
    print(json.dumps(dict(D=D, nlive=nlive, K=K, seed=seed, logz=float(r.logz[-1]), logzerr=float(r.logzerr[-1]),
                          niter=int(r.niter), ncall=int(np.sum(r.ncall)), truth=-D * np.log(2 * np.sqrt(np.pi)),
                          seconds=time.time() - t)), flush=True)
