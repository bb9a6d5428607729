import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, inputs
from oracle import friends_ref as F
from dynesty_amd import _lib
ctx=_lib.Context(0)
for kind in ("balls","cubes"):
  for name in ("two5","egg13"):
    pts=inputs.cloud(name); d=pts.shape[1]
    fr=F.friends_init(kind,d)
    f1,info=F.friends_update(fr,pts)
    f2,info2=F.friends_update(f1,pts)
    r=ctx.friends_update(pts,kind,axes_inv_prev=f1.axes_inv)
    ids=F.cluster_labels(pts,f1.am)
    print(kind,name,"ncl",r["nclusters"],info2["nclusters"],"rmax",r["rmax"],info2["rmax"], "cov/r2 err", np.abs(r["cov"]/r["rmax"]**2 - f2.cov/info2["rmax"]**2).max()/np.abs(f2.cov/info2["rmax"]**2).max(), np.bincount(ids))
    print(" am vs axes_inv^2", np.abs(f1.axes_inv@f1.axes_inv - f1.am).max()/np.abs(f1.am).max())
