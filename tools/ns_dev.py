import sys, time, json, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0)
def go(pname, runs, nlive, K, walks, bound='multi', **kw):
    prob=inputs.problem(pname)
    t=time.perf_counter()
    r=ctx.ns_ensemble(prob, runs, nlive, K, walks=walks, bound=bound, entropy=[21], **kw)
    kw.pop('sample',None); kw.pop('slices',None)
    dt=time.perf_counter()-t
    lz=r["logz"]
    print(json.dumps(dict(problem=pname,runs=runs,nlive=nlive,K=K,walks=walks,bound=bound,secs=round(dt,3),
        truth=prob.logz_truth, mean_logz=float(lz.mean()), std=float(lz.std(ddof=1)) if runs>1 else None,
        se=float(lz.std(ddof=1)/np.sqrt(runs)) if runs>1 else None, logzerr=float(r["logzerr"].mean()),
        niter=int(r["niter"].mean()), ncall=int(r["ncall"].mean()), nbound=float(r["nbound"].mean()),
        status=r["status"].tolist()[:8], nfills=r["nfills"], calls_per_s=float(r["ncall"].sum()/dt))))
go("G5", 8, 300, 64, 25)
go("C1", 8, 300, 64, 23, bound='single')
go("G5", 8, 300, 64, None, sample='rslice', slices=5)
go("G5", 8, 300, 64, None, sample='slice', slices=3)
go("C3", 16, 5000, 1024, None, sample='rslice', slices=5)
go("C2", 4, 2000, 512, 45)
go("C2", 64, 2000, 512, 45)
