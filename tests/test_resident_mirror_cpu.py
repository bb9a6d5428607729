"""The host mirror of the resident loop (tests/resident_mirror.py) on the ORACLE backend: every numerical step is then
the NumPy restatement pinned to the reference (oracle/, tests/oracle_backend.py), the control is the mirror's -- a
complete CPU restatement of csrc/ns.hip's loop.  Here (no GPU): ensembles of such runs reproduce the analytic ln Z, in
both forms of the forced bound update and for the three walkers; tests/test_gpu_resident_mirror.py holds the device loop
to the same runs event for event."""
import math

import numpy as np
import pytest

from oracle_backend import OracleBackend
from resident_mirror import mirror_run


@pytest.mark.parametrize("sample,K,bound,forced", [("rwalk", 1, "multi", "exact"), ("rwalk", 6, "single", "late"),
                                                   ("rslice", 4, "multi", "late"), ("unif", 8, "single", "late")])
def test_mirror_on_the_oracle_backend_gives_the_analytic_evidence(sample, K, bound, forced):
    from dynesty_amd import problems
    prob = problems.gauss_corr(4, 0.3, 5.0, "corr4")
    be = OracleBackend(canon=True)
    steps = dict(rwalk=15, rslice=5, unif=1)[sample]
    kw = dict(enlarge=1.0, bootstrap=0) if sample == "unif" else {}
    lz, forced_seen = [], 0
    for run in range(12):
        m = mirror_run(be, prob, 60, K, steps, bound, [11, K], run, 0.5, forced=forced, sample=sample, **kw)
        assert m["done"]
        lz.append(m["logz"])
        forced_seen += len(m["forced_fills"])
    lz = np.array(lz)
    se = lz.std(ddof=1) / math.sqrt(len(lz))
    # (dlogz = 0.5 leaves up to ~0.1 of the evidence in the final live points' tail estimate; 60 live points: sigma ~ 0.3)
    assert abs(lz.mean() - prob.logz_truth) < 4.0 * se + 0.15, (lz.mean(), prob.logz_truth, se)
    if sample == "rwalk" and K > 1:
        assert forced_seen > 0  # (start points outside the bound occurred: the forced update was walked)


@pytest.mark.parametrize("case", ["np16_rslice_K1", "np16_rslice_K32"])
def test_loop_protocol_against_800_real_reference_runs(case):
    """The resident loop's protocol (this mirror, oracle numerics) against ensembles of 800 REAL reference runs each
    (tests/golden/shape_logz_ref.json; tools/ref_shape_runs.py): the C4 family at 16-D -- Normal prior, iid Normal
    likelihood, nlive 300, single ellipsoid, rslice x 19 -- serial and with a queue of 32.  With 800 mirror runs each
    (tools/queue_effect_mirror.py, profiles/r04/queue_effect_cpu.json) the two agree to 0.002 +- 0.004 in ln Z, 0.1 % in
    iterations, 0.2 % in likelihood calls and to the bound-update count, at both queue sizes; here 16 / 10 runs (a run at K = 32 takes several seconds inside pytest)."""
    import json
    import os
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    assert ref["n"] >= 700
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    be = OracleBackend(canon=True)
    runs = [mirror_run(be, prob, c["nlive"], c["K"], c["slices"], c["bound"], [77, 16], r, c["dlogz"], sample=c["sample"])
            for r in range(16 if c["K"] == 1 else 10)]
    lz = np.array([m["logz"] for m in runs])
    se = math.hypot(lz.std(ddof=1) / math.sqrt(len(lz)), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 4.0 * se, (lz.mean(), ref["mean"], se)
    assert abs(np.mean([m["niter"] for m in runs]) / ref["mean_niter"] - 1) < 0.01
    assert abs(np.mean([m["ncall"] for m in runs]) / ref["mean_ncall"] - 1) < 0.02
    assert abs(np.mean([m["nbound"] for m in runs]) - (ref["mean_nbound"] - 1)) < 1.0
