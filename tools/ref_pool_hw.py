#!/usr/bin/env python
"""The reference's OWN parallel path on the host cores of the MI355X box (BASELINE.md section 3:
"dynesty.pool.Pool(ncores) with queue_size=ncores"): the real dynesty.NestedSampler(bound='multi', sample='rwalk',
walks=45, nlive=2000) on BASELINE config C2 with pool=dynesty.pool.Pool(ncores), queue_size=ncores
(pool.py:14-48, 148-173), timed over a bounded window after its first bound update, for each ncores given.

dynesty is not installed on the GPU box: run through a scratch copy that is never committed --

    bash tools/stage_reference.sh 'python tools/ref_pool_hw.py gpurun_out/refpool 32,128,256 40'

Writes reference_pool_on_gpu_box.json into the directory given (copied to profiles/rNN/ by hand; bench.py carries it
as cpu_baseline.reference_pool)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/refpool"
    cores = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "32,128").split(",")]
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
    os.makedirs(out_dir, exist_ok=True)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    import refshim
    dynesty = refshim.import_reference()
    import dynesty.pool as dpool
    from dynesty_amd import problems
    import bench
    prob = bench.c2_problem()
    nd, nlive, walks = prob.ndim, 2000, 45
    legs = []
    for nc in cores:
        nc = min(nc, os.cpu_count() or 1)
        t_spawn = time.perf_counter()
        with dpool.Pool(nc, prob.loglikelihood, prob.prior_transform) as pool:
            t_spawn = time.perf_counter() - t_spawn
            s = dynesty.NestedSampler(pool.loglike, pool.prior_transform, nd, nlive=nlive, bound='multi',
                                      sample='rwalk', walks=walks, pool=pool, queue_size=nc,
                                      rstate=np.random.default_rng(3))
            t0 = time.perf_counter()
            tb = ncall_b = it_b = None
            it = 0
            for it, _ in enumerate(s.sample(dlogz=0.01)):
                now = time.perf_counter()
                if tb is None and s.bound_list and len(s.bound_list) > 1:
                    tb, ncall_b, it_b = now, s.ncall, it
                if tb is not None and now - tb > budget:
                    break
                if now - t0 > 8 * budget:
                    break
            t1 = time.perf_counter()
        leg = dict(cores=nc, queue_size=nc, pool_start_seconds=t_spawn, seconds_before_first_bound=None if tb is None else tb - t0)
        if tb is not None:
            leg.update(seconds=t1 - tb, proposals=int(s.ncall - ncall_b), iterations=int(it - it_b),
                       proposals_per_s=float((s.ncall - ncall_b) / (t1 - tb)),
                       iterations_per_s=float((it - it_b) / (t1 - tb)), bound_updates=int(s.nbound))
        legs.append(leg)
        print(json.dumps(leg), flush=True)
    rec = dict(what="the reference's own parallel path: dynesty 3.0.0 (staged copy) NestedSampler(bound='multi', "
                    "sample='rwalk', walks=45, nlive=2000, pool=dynesty.pool.Pool(ncores), queue_size=ncores) on C2, "
                    f"the {budget:.0f} s after its first bound update; bound updates stay serial in the master "
                    "process (sampler.py:676-778), proposals go through pool.map with chunksize 1 (pool.py:148-157)",
               host=os.uname().nodename, cpu_count=os.cpu_count(), legs=legs)
    best = max((l for l in legs if "proposals_per_s" in l), key=lambda l: l["proposals_per_s"], default=None)
    if best:
        rec["best"] = dict(cores=best["cores"], proposals_per_s=best["proposals_per_s"])
    json.dump(rec, open(os.path.join(out_dir, "reference_pool_on_gpu_box.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":  # dynesty's Pool uses the spawn context: the workers re-import this module
    main()
