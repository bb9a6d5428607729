import sys, os, numpy as np, ctypes as C
os.environ["DYNHIP_LIB"]="/root/repo/dynesty_amd/libdynhip_timing.so"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0)
names=["mean","cov","regularize(all)","fmax","kmeans","partition","  jacobi","  sort_eigs","  copy/axes","  am","  km:vq","  km:sums","  km:update","  fast:ldl+inv","  fast:squaring","15"]
for cloud,multi in (("c2",True),("c2",False),("c3",True)):
    pts=inputs.cloud(cloud)
    ctx.rebuild(pts,multi=multi)
    out=(C.c_longlong*16)()
    ctx.lib.dh_rebuild_timing(out,1)
    ctx.rebuild(pts,multi=multi)
    ctx.lib.dh_rebuild_timing(out,1)
    tot=sum(out[:6])
    print(cloud,"multi" if multi else "single",{n:f"{out[i]/1e5:.1f}" for i,n in enumerate(names)},"total(1e5 ticks)",tot/1e5)
