import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import inputs
from dynesty_amd import _lib
ctx=_lib.Context(0); lib=ctx.lib; h=ctx.handle
import ctypes as C
def bench(name, runs, mode=0, reps=5):
    pts=inputs.cloud(name); n,d=pts.shape
    allp=np.concatenate([pts[np.random.default_rng(r).permutation(n)] for r in range(runs)])
    d_p=ctx.to_device(allp); me=max(1,n//(2*d))
    d_ne=ctx.malloc(runs*4); d_st=ctx.malloc(runs*4); d_nn=ctx.malloc(runs*4)
    d_c=ctx.malloc(runs*me*d*8); d_cov=ctx.malloc(runs*me*d*d*8); d_am=ctx.malloc(runs*me*d*d*8); d_ax=ctx.malloc(runs*me*d*d*8)
    d_al=ctx.malloc(runs*me*d*8); d_lv=ctx.malloc(runs*me*8)
    def go():
        ctx._check(lib.dh_rebuild_batch_dev(h,runs,d_p,n,d,mode,me,d_ne,d_st,d_c,d_cov,d_am,d_ax,d_al,d_lv,None,d_nn))
    go(); ctx.sync()
    e0,e1=ctx.event(),ctx.event()
    ctx.record(e0)
    for _ in range(reps): go()
    ctx.record(e1); ms=ctx.elapsed_ms(e0,e1)/reps
    ne=ctx.from_device(d_ne,(runs,),np.int32); st=ctx.from_device(d_st,(runs,),np.int32); nn=ctx.from_device(d_nn,(runs,),np.int32)
    print(f"{name} runs={runs} mode={mode}: {ms:.3f} ms/launch -> {runs/ms*1e3:.0f} rebuilds/s  nells={ne[:4]} status={st[:2]} nodes={nn[:3]}")
for runs in (1,64,256,512):
    bench("c2",runs)
bench("c2",1,mode=1); bench("c2",256,mode=1)
bench("c3",1); bench("c3",64)
bench("g3",1); bench("two5",256)
