import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from dynesty_amd import nested, problems
for d, nlive, K, sampler, bound in [(50,1000,500,'rslice','single'),(50,1000,32,'rslice','single'),(50,1000,500,'rwalk','single'),(20,1000,250,'rslice','single'),(20,1000,250,'rwalk','multi'),(100,2000,1000,'rslice','single')]:
    prob = problems.gauss_normal_prior(d, f"N{d}")
    out=[]
    for s in range(3):
        t=time.perf_counter()
        r = nested.run_static(prob, nlive=nlive, bound=bound, sample=sampler, queue_size=K,
                              rstate=np.random.default_rng(100+s), dlogz=0.01)
        out.append((r.logz, r.logzerr, r.niter, r.ncall, time.perf_counter()-t))
    lz=np.array([o[0] for o in out])
    print(json.dumps(dict(d=d,nlive=nlive,K=K,sampler=sampler,bound=bound,truth=prob.logz_truth,mean=lz.mean(),dev=lz.mean()-prob.logz_truth,logzerr=out[0][1],runs=[round(x,3) for x in lz],secs=round(out[0][4],2))))
