import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from dynesty_amd import nested, problems
prob = problems.gauss_normal_prior(48, "C4")
for bound, sample in (("multi", "rslice"), ("multi", "rwalk")):
    t = time.perf_counter()
    try:
        r = nested.run_static(prob, nlive=400, bound=bound, sample=sample, queue_size=64,
                              rstate=np.random.default_rng(5), dlogz=0.5, maxiter=None)
        print(json.dumps(dict(bound=bound, sample=sample, logz=r.logz, err=r.logzerr, truth=prob.logz_truth, niter=r.niter,
                              ncall=r.ncall, nbound=r.nbound, s=time.perf_counter() - t)))
    except Exception as e:
        print(bound, sample, "FAILED", type(e).__name__, str(e)[:300])
