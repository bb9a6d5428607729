"""The two forms of the forced bound update in the resident loop's PROTOCOL, on the CPU: runs of the host mirror
(tests/resident_mirror.py) on the oracle backend -- 13-D correlated Normal, nlive 100, MultiEllipsoid, rwalk x 30, queue of
16, where start points outside the bound are frequent -- one JSON line per run:
python tools/forced_forms_mirror.py late|exact first_run last_run"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_backend import OracleBackend  # noqa: E402
from resident_mirror import mirror_run  # noqa: E402
from dynesty_amd import problems  # noqa: E402

prob = problems.gauss_corr(13, 0.3, 5.0, "corr13")
be = OracleBackend(canon=True)
form, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for run in range(r0, r1):
    t = time.time()
    m = mirror_run(be, prob, 100, 16, 30, "multi", [55, 13], run, 0.1, forced=form)
    print(json.dumps(dict(form=form, run=run, logz=m["logz"], niter=m["niter"], ncall=m["ncall"], nbound=m["nbound"],
                          nforced=len(m["forced_fills"]), secs=round(time.time() - t, 1), truth=prob.logz_truth)), flush=True)
