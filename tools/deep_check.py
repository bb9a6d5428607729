import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from dynesty_amd import _lib
ctx=_lib.Context(0)
g=np.load('/root/repo/tests/golden/livesets.npz')
F=("ctrs","covs","ams","axes","axlens","logvol_ells")
sets=[g[f"C3/{i}/live_u"] for i in range(3)]
def run(env, many):
    old={k:os.environ.get(k) for k in ("DH_DEEP","DH_DEEP_FROM")}
    os.environ.update(env)
    try:
        if many: return ctx.rebuild_many(sets*4, multi=True)
        return [ctx.rebuild(s, multi=True) for s in sets]
    finally:
        for k,v in old.items():
            if v is None: os.environ.pop(k,None)
            else: os.environ[k]=v
for many in (False, True):
    a=run({"DH_DEEP":"0"}, many); b=run({}, many)
    for i,(x,y) in enumerate(zip(a,b)):
        same = x["nells"]==y["nells"] and all(np.array_equal(x[k][:x["nells"]],y[k][:y["nells"]]) for k in F)
        print("many" if many else "single", i, x["nells"], y["nells"], x.get("nnodes"), y.get("nnodes"), "identical" if same else "DIFFERENT",
              "" if same else max(np.abs(x[k][:x["nells"]]-y[k][:x["nells"]]).max() for k in F if x["nells"]==y["nells"]))
