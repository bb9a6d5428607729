"""Two processes on ONE GPU.  The rebuild's workgroups meet at spin waits whose
co-residency is derived from an occupancy query of a device the library assumes to be its own; a second process can hold
workgroup slots that sizing counts on.  What must hold then: a rebuild either returns the bits it returns alone, or fails
CLEANLY -- the bounded spin gives up, the run's status is DH_ERR_HIP and the call raises -- and nothing hangs.  Two
workers rebuild the same 96 live sets (more runs x parts than one chunk holds, so the chunked root and the chunked k-means
levels are in play) twenty times each, at the same time; every successful call is compared with the parent's own result."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from dynesty_amd import _lib
ref = np.load(sys.argv[2])
sets = [ref["set%d" % i] for i in range(int(ref["nsets"]))]
ctx = _lib.Context(0)
ok = fail = bad = 0
msgs = []
open(sys.argv[3] + ".ready", "w").write("1")
while not all(__import__("os").path.exists(p + ".ready") for p in sys.argv[4:]):
    time.sleep(0.01)
t0 = time.time()
for rep in range(20):
    try:
        out = ctx.rebuild_many(sets, multi=True)
    except Exception as e:  # the clean failure: DH_ERR_HIP of a starved run
        fail += 1
        msgs.append(str(e)[:200])
        continue
    ok += 1
    for i, o in enumerate(out):
        for k in ("ctrs", "covs", "ams", "axes", "axlens", "logvol_ells"):
            if not np.array_equal(np.asarray(o[k]), ref["%s%d" % (k, i)]):
                bad += 1
print(json.dumps(dict(ok=ok, fail=fail, bad=bad, secs=round(time.time() - t0, 3), msgs=msgs[:3])))
"""


def test_two_processes_on_one_gpu_give_the_same_bits_or_fail_cleanly(tmp_path):
    from dynesty_amd import _lib
    base = inputs.cloud("c2")
    sets = [base[np.random.default_rng(r).permutation(len(base))] for r in range(96)]
    ctx = _lib.Context(0)
    solo = ctx.rebuild_many(sets, multi=True)
    t0 = time.time()
    for _ in range(5):
        ctx.rebuild_many(sets, multi=True)
    solo_secs = (time.time() - t0) / 5
    del ctx
    arrays = dict(nsets=len(sets))
    for i, (s, o) in enumerate(zip(sets, solo)):
        arrays["set%d" % i] = s
        for k in ("ctrs", "covs", "ams", "axes", "axlens", "logvol_ells"):
            arrays["%s%d" % (k, i)] = np.asarray(o[k])
    ref = str(tmp_path / "ref.npz")
    np.savez(ref, **arrays)
    marks = [str(tmp_path / ("w%d" % i)) for i in range(2)]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, ref, marks[i]] + marks, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for i in range(2)]
    recs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a worker hung: the bounded spin waits did not give up")
        assert p.returncode == 0, err[-2000:]
        recs.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1]))
    for r in recs:
        assert r["ok"] + r["fail"] == 20
        assert r["bad"] == 0, recs           # never wrong bits
        for m in r["msgs"]:
            assert "-5" in m or "HIP" in m, m  # the spin limit's code, nothing else
    assert sum(r["ok"] for r in recs) > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_processes.json"), "w") as f:
        json.dump(dict(workers=recs, solo_seconds_per_call=round(solo_secs, 4), sets=96, calls_per_worker=20), f)
