#!/usr/bin/env python
"""40-D correlated Normal (rho 0.3, prior +-5), nlive 333, multi / rwalk x 60: the resident loop at K = 1 and K = 64 in both
RNG modes, to set beside the real reference (K = 1: -90.3 +- 0.3 over 4 runs; truth -92.10)."""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynesty_amd import _lib, problems
ctx = _lib.Context(0)
prob = problems.gauss_corr(40, 0.3, 5.0, "c40")
for K in (1, 64):
    for rng in ("pcg64", "philox"):
        for bound in ("multi", "single"):
            r = ctx.ns_ensemble(prob, 8, 333, K, bound=bound, sample="rwalk", walks=60, dlogz=0.5, entropy=[K, 5], rng=rng)
            print(json.dumps(dict(K=K, rng=rng, bound=bound, logz=round(float(r["logz"].mean()), 3),
                                  se=round(float(r["logz"].std(ddof=1) / np.sqrt(8)), 3), niter=int(r["niter"].mean()),
                                  ncall=int(r["ncall"].mean()), nbound=float(r["nbound"].mean()), status=r["status"].tolist())), flush=True)
