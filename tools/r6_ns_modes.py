"""64 full C2 runs through the resident loop in one RNG mode: python tools/r6_ns_modes.py pcg64|philox [runs] [K] [reps]"""
import os, sys, time, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from dynesty_amd import _lib
rng = sys.argv[1] if len(sys.argv) > 1 else "pcg64"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = _lib.Context(0)
prob = inputs.problem("C2")
for rep in range(reps):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 2000, K, walks=45, bound='multi', entropy=[21 + rep], rng=rng)
    dt = time.perf_counter() - t
    lz = r["logz"]
    print(json.dumps(dict(rng=rng, runs=runs, K=K, secs=round(dt, 4), mean_logz=float(lz.mean()),
                          se=float(lz.std(ddof=1) / np.sqrt(runs)), niter=int(r["niter"].mean()), ncall=int(r["ncall"].mean()),
                          nbound=float(r["nbound"].mean()), nfills=int(r["nfills"]), calls_per_s=float(r["ncall"].sum() / dt))))
