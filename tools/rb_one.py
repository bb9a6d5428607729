import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0)
name=sys.argv[1] if len(sys.argv)>1 else "c2"
pts=inputs.cloud(name)
for _ in range(3): r=ctx.rebuild(pts,multi=True)
print(r["nells"], r["nnodes"])
