#!/usr/bin/env python
"""Record the backend calls of REAL dynesty runs through the drop-in classes (build container only:
needs /root/reference) -> tests/golden/tapb_*.npz, replayed on the GPU by tests/test_gpu_tapb_replay.py.

  c1     BASELINE C1 (3-D Gaussian, single bound, uniform sampler, bootstrap 5), whole run, queue 32
  c2s    C2 settings at nlive 400, queue 64, 3500 iterations (the run of tests/test_same_seed_e2e.py)
  egg    C3's shape (2-D eggbox, multi / rslice) at nlive 500, queue 50, 2500 iterations
  c2     BASELINE C2 at full size (nlive 2000, queue 512): the bounded phase's first 10 queue fills
         (arguments of the device calls only -- the tap-B timing case)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refshim  # noqa: E402
import inputs  # noqa: E402
import tapb  # noqa: E402

dyn = refshim.import_reference()
from dynesty_amd import backend, dropin  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402


def record(tag, prob, nlive, K, seed, maxiter, make, stop_after_fills=None):
    rec = tapb.RecordingBackend(OracleBackend(canon=True))
    backend.set_backend(rec)

    class Stop(Exception):
        pass
    try:
        bnd, smp = make()
        pool = dropin.HipBatchPool(queue_size=K)
        if stop_after_fills:
            inner_map = pool.map
            state = {"n": 0}

            def counting(func, it):
                r = inner_map(func, it)
                if getattr(func, '_dynhip_batch', None) is not None:
                    state["n"] += 1
                    if state["n"] >= stop_after_fills:
                        raise Stop
                return r
            pool.map = counting
        s = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive, bound=bnd,
                              sample=smp, pool=pool, queue_size=K, rstate=np.random.default_rng(seed))
        try:
            s.run_nested(dlogz=0.01, maxiter=maxiter, print_progress=False)
        except (Stop, RuntimeError) as exc:
            if not isinstance(exc, Stop) and "Stop" not in repr(exc.__cause__):
                raise
    finally:
        backend.set_backend(None)
    names = {}
    for n, _, _ in rec.calls:
        names[n] = names.get(n, 0) + 1
    out = os.path.join(ROOT, "tests", "golden", f"tapb_{tag}.npz")
    tapb.save_trace(out, rec.calls, meta=dict(tag=tag, problem=prob.name, nlive=nlive, queue_size=K, seed=seed,
                                             niter=int(s.it), ncall=int(s.ncall), calls=names))
    print(tag, "iterations", s.it, "calls", names, "->", out, os.path.getsize(out) // 1024, "KiB")


which = sys.argv[1:] or ["c1", "c2s", "egg", "c2"]
if "c1" in which:
    p = inputs.problem("C1")
    record("c1", p, 300, 32, 2718, None, lambda: (dropin.HipEllipsoid(3), dropin.HipUniformBoundSampler(problem=p)))
if "c2s" in which:
    p = inputs.problem("C2")
    record("c2s", p, 400, 64, 314, 3500, lambda: (dropin.HipMultiEllipsoid(25), dropin.HipRWalkSampler(problem=p, walks=45)))
if "egg" in which:
    p = inputs.problem("C3")
    record("egg", p, 500, 50, 99, 2500, lambda: (dropin.HipMultiEllipsoid(2), dropin.HipRSliceSampler(problem=p, slices=5)))
if "c2" in which:
    p = inputs.problem("C2")
    record("c2", p, 2000, 512, 21, None, lambda: (dropin.HipMultiEllipsoid(25), dropin.HipRWalkSampler(problem=p, walks=45)),
           stop_after_fills=10)
