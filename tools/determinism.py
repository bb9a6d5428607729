"""Stress: repeated rebuilds of captured C3 / C2 live sets must be bit-identical (env: DH_SPLIT_TP, DH_DEEP)."""
import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from dynesty_amd import _lib
ctx=_lib.Context(0)
g=np.load('/root/repo/tests/golden/livesets.npz')
F=("ctrs","covs","ams","axes","axlens","logvol_ells")
reps=int(sys.argv[1]) if len(sys.argv)>1 else 100
for name in ("C3","C2"):
    base=[g[f"{name}/{i}/live_u"] for i in range(3)]
    rng=np.random.default_rng(0)
    sets=[b[rng.permutation(len(b))] for b in base for _ in range(6)]
    ref=ctx.rebuild_many(sets, multi=True)
    bad=0
    for r in range(reps):
        got=ctx.rebuild_many(sets, multi=True)
        for i,(x,y) in enumerate(zip(ref,got)):
            if x["nells"]!=y["nells"] or any(not np.array_equal(x[k],y[k]) for k in F):
                bad+=1
                if bad<4:
                    print(name,"rep",r,"set",i,"nells",x["nells"],y["nells"], [float(np.abs(x[k][:min(x["nells"],y["nells"])]-y[k][:min(x["nells"],y["nells"])]).max()) for k in F])
    print(name, "mismatching results:", bad, "of", reps*len(sets), flush=True)
