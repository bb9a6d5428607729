import sys, numpy as np
sys.path.insert(0,'/root/repo')
from dynesty_amd import problems, _lib, nested, backend
ctx=_lib.default_context(0)
prob=problems.gauss_normal_prior(48,"C4")
r=ctx.ns_ensemble(prob, 8, 400, 64, bound='single', sample='rwalk', dlogz=0.05, entropy=[48], max_iter=60000)
print("device loop", r["logz"].mean(), r["logz"].std(ddof=1), r["niter"].mean(), r["ncall"].mean(), r["nbound"].mean(), r["eff"].mean())
out=[nested.run_static(prob, nlive=400, bound='single', sample='rwalk', queue_size=64, rstate=np.random.default_rng(s), dlogz=0.05) for s in range(8)]
z=np.array([o.logz for o in out]); print("host loop", z.mean(), z.std(ddof=1), np.mean([o.niter for o in out]), np.mean([o.ncall for o in out]), np.mean([o.nbound for o in out]), np.mean([o.scale for o in out]))
print("truth", prob.logz_truth)
