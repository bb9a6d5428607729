"""The slice samplers in the resident loop's PROTOCOL on the CPU (tests/resident_mirror.py on the oracle backend): 6-D
correlated Normal, nlive 100, single ellipsoid, queue of 8 -- one JSON line per run (profiles/r04/slice_protocol_cpu.json):
python tools/slice_protocol_mirror.py slice|rslice first_run last_run"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_backend import OracleBackend
from resident_mirror import mirror_run
from dynesty_amd import problems
prob = problems.gauss_corr(6, 0.3, 5.0, "corr6")
be = OracleBackend(canon=True)
sample, r0, r1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = dict(slice=3, rslice=9, unif=1)[sample]
kw = dict(enlarge=1.0, bootstrap=5) if sample == "unif" else {}
for run in range(r0, r1):
    m = mirror_run(be, prob, 100, 8, steps, "single", [66, 6], run, 0.1, sample=sample, **kw)
    print(json.dumps(dict(sample=sample, run=run, logz=m["logz"], niter=m["niter"], ncall=m["ncall"], nbound=m["nbound"])), flush=True)
