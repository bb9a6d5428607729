import sys, time, os
sys.path.insert(0, ".")
from dynesty_amd import problems, _lib
prob = problems.gauss_normal_prior(200, "C4")
ctx = _lib.Context(0)
ctx.ns_ensemble(prob, 2, 4000, 128, bound="single", sample="rslice", slices=203, entropy=[3], max_fills=8, max_iter=250000)
t=time.perf_counter()
r = ctx.ns_ensemble(prob, 16, 4000, 128, bound="single", sample="rslice", slices=203, entropy=[21], max_iter=250000)
print("WPW", os.environ.get("DH_WIDE_WPW"), "secs", round(time.perf_counter()-t,3), r["nfills"], r["logz"].mean(), (r["status"]==0).all())
