"""The forced bound update (sampler.py:484-489) in its three forms -- none (DH_NS_FORCE=0: diagnostic), one fill late
(the default of the resident loop), inside the fill (forced_exact=True, the reference's sequence) -- on the shape cases
whose ensembles the real reference ran (tests/golden/shape_logz_ref.json) and on C2 / C4 for the cost.
python tools/forced_exact_cmp.py [out.jsonl] [c4]"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402
from dynesty_amd import _lib, problems  # noqa: E402

out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "shape_logz_ref.json")))["cases"]


def emit(row):
    line = json.dumps(row)
    print(line, flush=True)
    if out:
        out.write(line + "\n")
        out.flush()


def forms(name, prob, runs, nlive, K, ref_row=None, **kw):
    for form in ("none", "late", "exact"):
        if form == "none":
            os.environ["DH_NS_FORCE"] = "0"
        else:
            os.environ.pop("DH_NS_FORCE", None)
        ctx = _lib.Context(0)
        ctx.ns_ensemble(prob, 2, nlive, K, entropy=[1], max_fills=8, **kw)  # (module load, allocations)
        t0 = time.perf_counter()
        r = ctx.ns_ensemble(prob, runs, nlive, K, entropy=[31, 7], forced_exact=form == "exact", **kw)
        dt = time.perf_counter() - t0
        lz = r["logz"][r["status"] == 0]
        row = dict(case=name, form=form, runs=runs, ok=int((r["status"] == 0).sum()), secs=round(dt, 3),
                   mean_logz=float(lz.mean()), se=float(lz.std(ddof=1) / math.sqrt(len(lz))),
                   nbound=float(r["nbound"].mean()), niter=float(r["niter"].mean()), ncall=float(r["ncall"].mean()),
                   nfills=int(r["nfills"]))
        if ref_row:
            # counting conventions side by side: the reference's Sampler.nbound starts at 1 --
            # its initial UnitCube is bound number one (sampler.py:416) and every update adds one (:673) --, the
            # resident loop's record counts the UPDATES: nbound(loop) corresponds to nbound(reference) - 1
            rn = ref_row.get("mean_nbound")
            row.update(ref_mean=ref_row["mean"], ref_se=ref_row["se"], ref_nbound_incl_unit_cube=rn,
                       ref_bound_updates=None if rn is None else rn - 1.0,
                       bound_updates_loop_minus_reference=None if rn is None else row["nbound"] - (rn - 1.0))
        emit(row)
        del ctx


if len(sys.argv) > 2 and sys.argv[2] == "c4":
    forms("C4", problems.gauss_normal_prior(200, "C4"), 16, 4000, 128, bound="single", sample="rslice", slices=203,
          max_iter=250000)
    sys.exit(0)
for case in ("rwalk25_K1", "rwalk44_multi", "rslice40_multi", "slice36_single", "rwalk13_multi", "multi2_tiny"):
    c = ref[case]["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge") if k in c}
    forms(case, prob, 64, c["nlive"], c["K"], ref_row=ref[case], bound=c["bound"], sample=c["sample"],
          dlogz=c.get("dlogz", 0.5), **kw)
forms("C2", inputs.problem("C2"), 64, 2000, 512, walks=45, bound="multi")
