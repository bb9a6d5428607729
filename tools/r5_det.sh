# determinism stress of the final code (repeat-and-compare): the bench shard in a loop + tools/determinism_all.py
mkdir -p gpurun_out/r5det
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/r5det/det.txt
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import bench
from dynesty_amd import _lib
ctx = _lib.Context(0)
for runs in (64, 160):
    sh = bench.Shard(ctx, bench.c2_problem(), runs=runs, nlive=2000, walks=45, seed=1000)
    sh.rebuild(enlarge=False); ctx.sync(); ref = sh.fetch_bound(); bad = 0
    for i in range(200):
        sh.rebuild(enlarge=False); ctx.sync(); b = sh.fetch_bound()
        bad += any(not np.array_equal(ref[k], b[k]) for k in ("nells", "ctrs", "covs", "ams", "axes", "logvols", "status"))
    print(f"bench shard {runs} runs: {bad} differing rebuilds of 200", flush=True)
PY
timeout 900 python tools/determinism_all.py 2>&1 | tee -a gpurun_out/r5det/det.txt
