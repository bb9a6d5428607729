import sys, os, numpy as np, ctypes as C
os.environ["DYNHIP_LIB"]="/root/repo/dynesty_amd/libdynhip_timing.so"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0)
names=["mean","cov","regularize(all)","fmax","kmeans","fmax:stage","  jacobi","  sort_eigs","  copy/axes","  am","  km:vq","  km:sums","  km:update","  fast:sweep-inv","  fast:squaring","  fast:eigvec"]
c2=inputs.cloud("c2")
for n in (2000, 500, 250, 130):
    pts=c2[:n]
    ctx.rebuild(pts,multi=True)
    out=(C.c_longlong*16)()
    ctx.lib.dh_rebuild_timing(out,1)
    r=ctx.rebuild(pts,multi=True)
    ctx.lib.dh_rebuild_timing(out,1)
    print(n,"nnodes",r["nnodes"],{nm.strip():round(out[i]/1e3,1) for i,nm in enumerate(names) if out[i]},"(1e3 cycles, workgroup 0 of every kernel)")
