"""Repeat-and-compare stress over the entry points that synchronise workgroups through memory or atomics."""
import os, sys, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import inputs
from dynesty_amd import _lib, problems
ctx=_lib.Context(0)
F=("ctrs","covs","ams","axes","axlens","logvol_ells")
def same(a,b): return a["nells"]==b["nells"] and all(np.array_equal(a[k],b[k]) for k in F)
# wide single rebuild (multi-workgroup block Jacobi with exchange barriers)
rng=np.random.default_rng(1)
for (n,d,reps) in ((4000,200,60),(1200,96,60),(900,64,60)):
    pts=0.5+0.05*rng.standard_normal((n,d))@ (np.eye(d)+0.1*rng.standard_normal((d,d)))
    ref=ctx.rebuild(pts,multi=False); bad=0
    for _ in range(reps): bad+= not same(ref, ctx.rebuild(pts,multi=False))
    print(f"wide single {n}x{d}: {bad} differing of {reps}", flush=True)
# wide multi
pts=np.concatenate([0.3+0.02*rng.standard_normal((700,48)), 0.7+0.02*rng.standard_normal((700,48))])
ref=ctx.rebuild(pts,multi=True); bad=0
for _ in range(40): bad+= not same(ref, ctx.rebuild(pts,multi=True))
print("wide multi 1400x48:", bad, "differing of 40", flush=True)
# cooperative root + ragged batches
c3=inputs.cloud("c3"); sets=[c3[rng.permutation(5000)[:m]] for m in (5000,4999,1234,700,257,256,255)]*6
ref=ctx.rebuild_many(sets,multi=True); bad=0
for _ in range(60):
    got=ctx.rebuild_many(sets,multi=True); bad+=sum(not same(a,b) for a,b in zip(ref,got))
print("ragged c3 batches:", bad, "differing of", 60*len(sets), flush=True)
# device NS loop
prob=inputs.problem("C2")
a=ctx.ns_ensemble(prob,16,2000,512,walks=45,entropy=[3]); bad=0
for _ in range(4):
    b=ctx.ns_ensemble(prob,16,2000,512,walks=45,entropy=[3]); bad+=int((a["logz"]!=b["logz"]).sum()+(a["ncall"]!=b["ncall"]).sum())
print("ns_ensemble C2 16 runs x 4 repeats: differing", bad, flush=True)
# friends update
from dynesty_amd import bounding
e=inputs.cloud("egg13")
r0=ctx.friends_update(e,'balls'); bad=0
for _ in range(50):
    r=ctx.friends_update(e,'balls'); bad+= any(not np.array_equal(r0[k],r[k]) for k in ("cov","am","axes","axes_inv"))
print("friends_update: differing", bad, "of 50")
