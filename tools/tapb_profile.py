"""cProfile of one tap-B run (the unmodified dynesty.NestedSampler over the drop-in classes, C2, K = 512) on the GPU box:
where pool.map and the host loop spend their time.  Needs the staged reference (tools/stage_reference.sh)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refshim
dynesty = refshim.import_reference()
import bench
from dynesty_amd import dropin
prob = bench.c2_problem()
nd, nlive, walks, K = prob.ndim, 2000, 45, 512
bound = dropin.HipMultiEllipsoid(nd)
pool = dropin.HipBatchPool(queue_size=K)
s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, nd, nlive=nlive, bound=bound,
                          sample=dropin.HipRWalkSampler(problem=prob, walks=walks), pool=pool, queue_size=K,
                          rstate=np.random.default_rng(5))
pr = cProfile.Profile()
pr.enable()
s.run_nested(dlogz=0.01, print_progress=False, maxiter=30000)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
