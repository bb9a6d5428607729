"""BASELINE C4 through the device-resident loop at several queue sizes: ensemble ln Z and time per run.
python tools/c4_ksweep.py runs K [K ...]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
ctx = _lib.Context(0)
prob = problems.gauss_normal_prior(200, "C4")
runs = int(sys.argv[1])
kw = dict(bound='single', sample='rslice', slices=203, dlogz=0.01, max_iter=250000)
ctx.ns_ensemble(prob, 1, 4000, 1000, entropy=[3], max_fills=2, **kw)
for K in map(int, sys.argv[2:]):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 4000, K, entropy=[21, K], **kw)
    dt = time.perf_counter() - t
    lz = r["logz"]
    print(json.dumps(dict(K=K, runs=runs, secs=round(dt, 2), secs_per_run=round(dt / runs, 3), logz=round(float(lz.mean()), 4),
                          se=round(float(lz.std(ddof=1) / np.sqrt(runs)), 4), niter=int(r["niter"].mean()),
                          ncall=int(r["ncall"].mean()), ok=bool((r["status"] == 0).all()))), flush=True)
