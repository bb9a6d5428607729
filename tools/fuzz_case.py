"""One shape of tools/shape_cases.json on the device in the four combinations of RNG mode and forced-update form:
python tools/fuzz_case.py <case> [runs]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
c = json.load(open(os.path.join(ROOT, "tools", "shape_cases.json")))[sys.argv[1]]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
ctx = _lib.Context(0)
kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge", "update_interval", "first_update") if k in c}
for rng in ("pcg64", "philox"):
    for exact in (False, True):
        r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                            entropy=[77, 1], rng=rng, forced_exact=exact, **kw)
        lz = r["logz"]
        print(json.dumps(dict(rng=rng, exact=exact, ok=bool((r["status"] == 0).all()), logz=round(float(lz.mean()), 3),
                              se=round(float(lz.std(ddof=1) / np.sqrt(runs)), 3), niter=float(r["niter"].mean()),
                              ncall=float(r["ncall"].mean()), nbound=float(r["nbound"].mean()), nfills=int(r["nfills"]))), flush=True)
