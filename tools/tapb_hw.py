#!/usr/bin/env python
"""Tap B on the hardware: the REAL dynesty.NestedSampler driving the
device through the drop-in surface -- bound=HipMultiEllipsoid, sample=HipRWalkSampler, pool=HipBatchPool(512) -- on
BASELINE config C2 to dlogz = 0.01, and beside it the reference's own CPU path on one host core of the same box
(bounded: proposals/s of the bounded phase, ms per MultiEllipsoid.update).

dynesty is not installed on the GPU box and the reference tree does not travel, so this is run by hand through a
scratch copy that is NEVER committed (.gitignore: _refstage/) and is deleted right after:

    mkdir -p _refstage && cp -r /root/reference/py _refstage/py
    gpurun --timeout 900 -- 'DYNESTY_REF_PY=$PWD/_refstage/py python tools/tapb_hw.py gpurun_out/tapb'
    rm -rf _refstage
    cp gpurun_out/tapb/*.json profiles/r04/

Outputs tapb_c2.json and reference_cpu_on_gpu_box.json into the directory given."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tapb"
budget_ref = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
os.makedirs(out_dir, exist_ok=True)

import refshim  # noqa: E402
dynesty = refshim.import_reference()
import bench  # noqa: E402
from dynesty_amd import dropin  # noqa: E402

prob = bench.c2_problem()
nd, nlive, walks, K = prob.ndim, 2000, 45, 512


class Timed:
    """wall time spent inside a callable, summed"""

    def __init__(self, fn):
        self.fn, self.t, self.n = fn, 0.0, 0

    def __call__(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return self.fn(*a, **k)
        finally:
            self.t += time.perf_counter() - t0
            self.n += 1


# ---- (i) the real NestedSampler over the drop-in classes, on the device ----
res = {}
for seed in (() if os.environ.get("TAPB_SKIP_DEVICE") else (5, 6)):
    bound = dropin.HipMultiEllipsoid(nd)
    pool = dropin.HipBatchPool(queue_size=K)
    pool.map = Timed(pool.map)
    bound.update = Timed(bound.update)
    s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, nd, nlive=nlive, bound=bound,
                              sample=dropin.HipRWalkSampler(problem=prob, walks=walks), pool=pool, queue_size=K,
                              rstate=np.random.default_rng(seed))
    t0 = time.perf_counter()
    s.run_nested(dlogz=0.01, print_progress=False)
    dt = time.perf_counter() - t0
    r = s.results
    ncall = int(np.sum(r.ncall))
    res[f"seed{seed}"] = dict(
        seconds=dt, niter=int(r.niter), ncall=ncall, logz=float(r.logz[-1]), logzerr=float(r.logzerr[-1]),
        iterations_per_s=r.niter / dt, proposals_per_s=ncall / dt,
        seconds_in_pool_map=pool.map.t, pool_map_calls=pool.map.n, seconds_in_bound_update=bound.update.t,
        bound_updates=bound.update.n, host_python_seconds=dt - pool.map.t - bound.update.t,
        proposals_per_s_device_side=ncall / max(pool.map.t, 1e-9))
tapb = dict(what="tap B on the MI355X box: unmodified dynesty.NestedSampler(bound=HipMultiEllipsoid, "
                 "sample=HipRWalkSampler(walks=45), pool=HipBatchPool(512), queue_size=512), C2 (25-D correlated "
                 "Normal, nlive 2000), dlogz 0.01; the run loop is the reference's serial host Python",
            reference_logz="tests/golden/c2_logz_ref.json (K = 512: -57.493 +/- 0.023, 32 runs)", runs=res)
if res:
    json.dump(tapb, open(os.path.join(out_dir, "tapb_c2.json"), "w"), indent=1)
    print(json.dumps(tapb, indent=1))

# ---- (ii) the reference's own CPU path on one host core of this box, bounded ----
from dynesty import bounding as db  # noqa: E402
rng = np.random.default_rng(11)
u0, loglstar = bench.make_shard(prob, 1, nlive, 1000)
live = u0[:nlive]
t_up = []
ell = db.MultiEllipsoid(nd)
for i in range(5):
    t0 = time.perf_counter()
    ell.update(live, rstate=rng)
    t_up.append(time.perf_counter() - t0)
# bounded phase of the reference itself: a fresh serial run, timed once its bound exists, until the budget is spent
s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, nd, nlive=nlive, bound='multi', sample='rwalk',
                          walks=walks, rstate=np.random.default_rng(3))
t0 = time.perf_counter()
ncall_b = it_b = None
tb = None
for it, results in enumerate(s.sample(dlogz=0.01)):
    if tb is None and s.bound_list and len(s.bound_list) > 1:  # first real bound built: bounded phase starts
        tb, ncall_b, it_b = time.perf_counter(), s.ncall, it
    if tb is not None and time.perf_counter() - tb > budget_ref:
        break
    if time.perf_counter() - t0 > 6 * budget_ref:
        break
t1 = time.perf_counter()
ref = dict(what="the reference's own CPU path (dynesty 3.0.0 from the staged copy), ONE host core of the GPU box",
           cores=1, host=os.uname().nodename, cpu_count=os.cpu_count(),
           multiellipsoid_update_ms=dict(median=float(np.median(t_up) * 1e3), all=[float(x * 1e3) for x in t_up],
                                         points="64 x ... the bench shard's first live set (2000 x 25)"),
           bounded_phase=None if tb is None else dict(
               seconds=t1 - tb, proposals=int(s.ncall - ncall_b), iterations=int(it - it_b),
               proposals_per_s=float((s.ncall - ncall_b) / (t1 - tb)), iterations_per_s=float((it - it_b) / (t1 - tb))),
           sample="serial NestedSampler(bound='multi', sample='rwalk', walks=45, nlive=2000) on C2, the "
                  f"{budget_ref:.0f} s after its first bound update")
json.dump(ref, open(os.path.join(out_dir, "reference_cpu_on_gpu_box.json"), "w"), indent=1)
print(json.dumps(ref, indent=1))
