import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0)
prob=inputs.problem("C3")
def go(env, fills):
    old={k:os.environ.get(k) for k in ("DH_DEEP",)}
    os.environ.update(env)
    try:
        return ctx.ns_ensemble(prob, 16, 5000, 1024, bound='multi', sample='rslice', slices=5, entropy=[22], max_fills=fills)
    finally:
        for k,v in old.items():
            if v is None: os.environ.pop(k,None)
            else: os.environ[k]=v
for fills in (8, 16, 24, 32, 64):
    a=go({"DH_DEEP":"0"}, fills); b=go({}, fills); c=go({}, fills)
    d=np.flatnonzero((a["logz"]!=b["logz"])|(a["ncall"]!=b["ncall"]))
    print(fills, "differing runs", d.tolist(), "status", a["status"].tolist()[:4], b["status"].tolist()[:4], "repeat identical", bool((b["logz"]==c["logz"]).all()),
          "nbound", a["nbound"][:6].tolist(), b["nbound"][:6].tolist())
