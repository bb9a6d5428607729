"""BASELINE config C1 through the device-resident loop with the reference's defaults for sample='unif' (bootstrap 5,
enlarge 1): usage ns_c1.py [runs] [K]"""
import os, sys, time, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = _lib.Context(0)
prob = problems.gauss_iid(3, 10.0, "C1")
for bound in ("single", "multi", "single"):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 500, K, bound=bound, sample='unif', entropy=[21], dlogz=0.01)
    dt = time.perf_counter() - t
    lz = r["logz"]
    print(json.dumps(dict(bound=bound, runs=runs, K=K, secs=round(dt, 3), mean_logz=float(lz.mean()),
                          se=float(lz.std(ddof=1) / np.sqrt(runs)), niter=int(r["niter"].mean()),
                          ncall=int(r["ncall"].mean()), nbound=float(r["nbound"].mean()), nfills=r["nfills"])))
