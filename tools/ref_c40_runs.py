"""The real reference on a 40-D correlated Normal with few live points (tests/golden/c40_logz_ref.json):
   python tools/ref_c40_runs.py <seed> [K [bound]]   (one run per call; K > 1 uses a SerialPool(K))"""
import sys, time, json
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import refshim
refshim.import_reference()
import numpy as np
import dynesty
from dynesty_amd import problems
prob = problems.gauss_corr(40, 0.3, 5.0, "c40")
seed = int(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
class SerialPool:
    def __init__(self, size):
        self.size = size
    def map(self, f, it):
        return list(map(f, it))

t = time.time()
s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 40, nlive=333, bound=(sys.argv[3] if len(sys.argv) > 3 else 'multi'), sample='rwalk', walks=60, **(dict(pool=SerialPool(K), queue_size=K) if K > 1 else {}),
                          rstate=np.random.default_rng(seed))
s.run_nested(dlogz=0.5, print_progress=False)
r = s.results
print(json.dumps(dict(seed=seed, K=K, bound=(sys.argv[3] if len(sys.argv) > 3 else 'multi'), nbound=int(r['bound_iter'].max()) if 'bound_iter' in r.keys() else -1, logz=float(r.logz[-1]), err=float(r.logzerr[-1]), niter=int(r.niter), ncall=int(sum(r.ncall)), secs=time.time() - t, truth=prob.logz_truth)), flush=True)
