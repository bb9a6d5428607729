"""Reference run behind tests/golden/rslice_bias_ref.json: the REAL dynesty (from /root/reference/py,
build container only) on the C4 family at dimension D -- usage: ref_rslice_bias.py D nlive seed."""
import sys, time, json
sys.path.insert(0, "/root/repo/tests"); import refshim; refshim.import_reference()
import numpy as np
from scipy.special import ndtri
import dynesty
D = int(sys.argv[1]); nlive = int(sys.argv[2]); seed = int(sys.argv[3])
c = -0.5 * D * np.log(2 * np.pi)
def loglike(x): return c - 0.5 * np.dot(x, x)
def ptform(u): return ndtri(u)
t = time.time()
s = dynesty.NestedSampler(loglike, ptform, D, nlive=nlive, bound='single', sample='rslice',
                          rstate=np.random.default_rng(seed))
s.run_nested(dlogz=0.01, print_progress=False)
r = s.results
print(json.dumps(dict(D=D, nlive=nlive, seed=seed, logz=float(r.logz[-1]), logzerr=float(r.logzerr[-1]),
                      niter=int(r.niter), ncall=int(np.sum(r.ncall)), truth=-D * np.log(2 * np.sqrt(np.pi)),
                      seconds=time.time() - t)))
