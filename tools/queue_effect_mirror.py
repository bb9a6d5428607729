"""Queue-size effect of the resident loop's PROTOCOL on the CPU: runs of the host mirror (tests/resident_mirror.py) on the
oracle backend -- the C4 family at 16-D (Normal prior, iid Normal likelihood, nlive 300, single ellipsoid, rslice) -- at a
queue size K, one JSON line per run:  python tools/queue_effect_mirror.py K first_run last_run"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_backend import OracleBackend
from resident_mirror import mirror_run
from dynesty_amd import problems
D, N = 16, 300
prob = problems.gauss_normal_prior(D, "np16")
be = OracleBackend(canon=True)
K = int(sys.argv[1]); r0 = int(sys.argv[2]); r1 = int(sys.argv[3])
for run in range(r0, r1):
    t = time.time()
    m = mirror_run(be, prob, N, K, 3 + D, "single", [77, 16], run, 0.1, sample="rslice")
    print(json.dumps(dict(K=K, run=run, logz=m["logz"], niter=m["niter"], ncall=m["ncall"], nbound=m["nbound"], secs=round(time.time() - t, 1), truth=prob.logz_truth)), flush=True)
