"""Device side of the shape check: the resident loop on the shapes of tools/shape_cases.json, 16 runs each."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
ctx = _lib.Context(0)
NRUN = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY = sys.argv[2:]
for name, c in json.load(open(os.path.join(ROOT, "tools", os.environ.get("SHAPE_CASES", "shape_cases.json")))).items():
    if ONLY and name not in ONLY:
        continue
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge") if k in c}
    for rng in ("pcg64", "philox"):
        try:
            r = ctx.ns_ensemble(prob, (NRUN if NRUN < 40 else (8 if c["prob"][1] >= 36 else NRUN)), c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                                entropy=[7, len(name)], rng=rng, **kw)
            lz = r["logz"]
            print(json.dumps(dict(case=name, rng=rng, logz=round(float(lz.mean()), 3), se=round(float(lz.std(ddof=1) / np.sqrt(len(lz))), 3),
                                  niter=int(r["niter"].mean()), ncall=int(r["ncall"].mean()), nbound=float(r["nbound"].mean()),
                                  ok=bool((r["status"] == 0).all()), truth=prob.logz_truth)), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps(dict(case=name, rng=rng, error=repr(ex)[:200])), flush=True)
