"""Experiment: the C5 shard as G concurrent device-resident ensembles (one context / stream / host thread each)."""
import os, sys, time, threading, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from dynesty_amd import _lib
prob = inputs.problem("C2")
for G in (1, 2, 4):
    ctxs = [_lib.Context(0) for _ in range(G)]
    per = 64 // G
    for c in ctxs:
        c.ns_ensemble(prob, per, 2000, 512, walks=45, entropy=[3], max_fills=4)
    for sync in (False, True):
        out = [None] * G
        def work(g):
            out[g] = ctxs[g].ns_ensemble(prob, per, 2000, 512, walks=45, entropy=[21], first_run=g * per, rebuild_sync=sync)
        t = time.perf_counter()
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        [x.start() for x in th]; [x.join() for x in th]
        dt = time.perf_counter() - t
        lz = np.concatenate([o["logz"] for o in out])
        print(f"groups={G} rebuild_sync={int(sync)}: {dt:.3f} s  lnZ {lz.mean():.4f} +- {lz.std(ddof=1)/8:.4f}  status_ok={all((o['status']==0).all() for o in out)}", flush=True)
