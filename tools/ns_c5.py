"""One BASELINE-C5 shard on the device: R full C2 runs through dh_ns_ensemble.  usage: ns_c5.py [runs] [K]"""
import os, sys, time, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from dynesty_amd import _lib
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = _lib.Context(0)
prob = inputs.problem("C2")
for rep in range(1 if (len(sys.argv) > 3 and sys.argv[3] == "once") else 4):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, runs, 2000, K, walks=45, bound='multi', entropy=[21 + rep // 2], rebuild_sync=bool(rep % 2))
    dt = time.perf_counter() - t
    lz = r["logz"]
    print(json.dumps(dict(rebuild_sync=bool(rep % 2), runs=runs, K=K, secs=round(dt, 3), mean_logz=float(lz.mean()),
                          se=float(lz.std(ddof=1) / np.sqrt(runs)), niter=int(r["niter"].mean()),
                          ncall=int(r["ncall"].mean()), nbound=float(r["nbound"].mean()), nfills=r["nfills"],
                          calls_per_s=float(r["ncall"].sum() / dt))))
