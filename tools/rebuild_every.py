"""Loop time against rebuild_every (bounds built every n-th fill, due runs wait): usage rebuild_every.py [c4runs]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
ctx = _lib.Context(0)
c4runs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = [
    ("C2", problems.gauss_corr(25, 0.4, 5.0, "C2"), 64, 2000, 512, dict(bound='multi', sample='rwalk', walks=45)),
    ("C3", problems.eggbox(2, name="C3"), 16, 5000, 1024, dict(bound='multi', sample='rslice', slices=5)),
    ("C1", problems.gauss_iid(3, 10.0, "C1"), 64, 500, 64, dict(bound='single', sample='unif')),
]
if c4runs:
    cases.append(("C4", problems.gauss_normal_prior(200, "C4"), c4runs, 4000, 128,
                  dict(bound='single', sample='rslice', slices=203, max_iter=250000)))
for name, prob, runs, nlive, K, kw in cases:
    ctx.ns_ensemble(prob, 1, nlive, K, entropy=[3], max_fills=2, dlogz=0.01, **kw)
    for every in [int(x) for x in os.environ.get('EVERY', '1,2,3,4,0').split(',')]:
        t = time.perf_counter()
        r = ctx.ns_ensemble(prob, runs, nlive, K, entropy=[21, K], dlogz=0.01, rebuild_every=every, **kw)
        dt = time.perf_counter() - t
        lz = r["logz"]
        print(json.dumps(dict(config=name, rebuild_every=every, runs=runs, K=K, secs=round(dt, 3), logz=float(lz.mean()),
                              nfills=int(r["nfills"]), nbound=float(r["nbound"].mean()), ok=bool((r["status"] == 0).all()))), flush=True)
