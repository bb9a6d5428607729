import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import inputs
from dynesty_amd import _lib
ctx = _lib.Context(0)
prob = inputs.problem("G5")
for (n, k) in ((20000, 1024), (30000, 256)):
    t = time.perf_counter()
    r = ctx.ns_ensemble(prob, 2, n, k, walks=25, bound="multi", entropy=[20, 0, 0, 0], dlogz=0.5, max_iter=900000)
    print(n, k, round(time.perf_counter() - t, 3), r["niter"], r["logz"], r["nfills"], flush=True)
