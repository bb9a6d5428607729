"""One real-dynesty run of a shape given on the command line (JSON) -- the reference side of tools/shape_check.py:
   python tools/ref_shape_runs.py '{"prob": ["gauss_corr", 25, 0.3, 5.0], "nlive": 60, "K": 257, "bound": "single",
                                     "sample": "rwalk", "dlogz": 0.5}' <seed>"""
import sys, time, json
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import refshim
refshim.import_reference()
import numpy as np
import dynesty
from dynesty_amd import problems


class SerialPool:
    def __init__(self, size):
        self.size = size

    def map(self, f, it):
        return list(map(f, it))


c = json.loads(sys.argv[1])
seed = int(sys.argv[2])
prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
kw = dict(nlive=c["nlive"], bound=c["bound"], sample=c["sample"], rstate=np.random.default_rng(seed))
for k in ("walks", "slices", "bootstrap", "enlarge", "update_interval", "first_update", "periodic", "reflective"):
    if k in c:
        kw[k] = c[k]
# run_nested's own options (round 4: the resident loop takes them too)
rkw = {k: c[k] for k in ("maxiter", "maxcall", "logl_max", "add_live") if k in c}
if c["K"] > 1:
    kw.update(pool=SerialPool(c["K"]), queue_size=c["K"])
t = time.time()
s = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, **kw)
s.run_nested(dlogz=c.get("dlogz", 0.5), print_progress=False, **rkw)
r = s.results
print(json.dumps(dict(seed=seed, logz=float(r.logz[-1]), err=float(r.logzerr[-1]), niter=int(r.niter), ncall=int(sum(r.ncall)),
                      nbound=int(s.nbound), secs=time.time() - t, truth=prob.logz_truth)), flush=True)
