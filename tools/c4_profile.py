#!/usr/bin/env python
"""Where the wall time of the C4 run goes on the host side (cProfile, top entries)."""
import cProfile, pstats, sys, io
import numpy as np
sys.path.insert(0, ".")
from dynesty_amd import nested, problems
prob = problems.gauss_normal_prior(200, "C4")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nested.run_static(prob, nlive=4000, bound='single', sample='rslice', slices=203, queue_size=K,
                  rstate=np.random.default_rng(3), dlogz=0.01, maxiter=2000)  # warm-up
pr = cProfile.Profile()
pr.enable()
r = nested.run_static(prob, nlive=4000, bound='single', sample='rslice', slices=203, queue_size=K,
                      rstate=np.random.default_rng(21), dlogz=0.01)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
print("niter", r.niter, "ncall", r.ncall, "nbound", r.nbound)
