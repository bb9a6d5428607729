"""C3 ensemble (eggbox 2-D, nlive 5000, multi/rslice): python tools/ns_c3.py [runs]"""
import os, sys, time, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from dynesty_amd import _lib
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = _lib.Context(0)
prob = inputs.problem("C3")
for rep in range(2):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 5000, 1024, bound='multi', sample='rslice', slices=5, entropy=[21 + rep])
    dt = time.perf_counter() - t
    lz = r["logz"]
    print(json.dumps(dict(runs=runs, secs=round(dt, 3), mean_logz=float(lz.mean()), se=float(lz.std(ddof=1) / np.sqrt(runs)),
                          truth=prob.logz_truth, niter=int(r["niter"].mean()), ncall=int(r["ncall"].mean()),
                          nbound=float(r["nbound"].mean()), nfills=r["nfills"])))
