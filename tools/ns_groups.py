"""C5 shard (64 C2 runs) as G concurrent groups, one Context (stream + workspaces) and host thread per group.
usage: ns_groups.py [runs] [K] [groups...]"""
import sys, time, json, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import inputs
from dynesty_amd import _lib
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
groups = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
prob = inputs.problem("C2")
ctxs = [_lib.Context(0) for _ in range(max(groups))]
for c in ctxs:  # warm up each context (problem upload, workspaces)
    c.ns_ensemble(prob, 2, 2000, K, walks=45, bound='multi', entropy=[1], max_fills=3)
for G in groups:
    for sync in (False, True):
        out = [None] * G
        per = runs // G
        def work(g):
            out[g] = ctxs[g].ns_ensemble(prob, per, 2000, K, walks=45, bound='multi', entropy=[21],
                                         first_run=g * per, rebuild_sync=sync)
        t = time.perf_counter()
        th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        [x.start() for x in th]; [x.join() for x in th]
        dt = time.perf_counter() - t
        lz = np.concatenate([o["logz"] for o in out]); nc = sum(int(o["ncall"].sum()) for o in out)
        print(json.dumps(dict(groups=G, rebuild_sync=sync, runs=per * G, secs=round(dt, 3), mean_logz=float(lz.mean()),
                              se=float(lz.std(ddof=1) / np.sqrt(len(lz))), calls_per_s=nc / dt)))
