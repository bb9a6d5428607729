"""Fold ensembles written by tools/ref_shape_runs.py (one JSON line per run, files <case>.jsonl in DIR) into
tests/golden/shape_logz_ref.json:  python tools/ref_shape_collect.py DIR case [case ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "shape_logz_ref.json")
out = json.load(open(path))
cases = json.load(open(os.path.join(ROOT, "tools", "shape_cases.json")))
src = sys.argv[1]
for name in sys.argv[2:]:
    rows = [json.loads(line) for line in open(os.path.join(src, name + ".jsonl")) if line.strip()]
    lz = np.array([r["logz"] for r in rows])
    out["cases"][name] = dict(config=cases[name], n=len(rows), mean=float(lz.mean()),
                              se=float(lz.std(ddof=1) / np.sqrt(len(rows))),
                              mean_niter=float(np.mean([r["niter"] for r in rows])),
                              mean_ncall=float(np.mean([r["ncall"] for r in rows])),
                              mean_nbound=float(np.mean([r["nbound"] for r in rows])), truth=rows[0]["truth"],
                              logz=[round(float(x), 6) for x in lz], seeds=[int(r["seed"]) for r in rows],
                              ncalls=[int(r["ncall"]) for r in rows])
    print(name, {k: v for k, v in out["cases"][name].items() if k not in ("logz", "seeds", "config")})
json.dump(out, open(path, "w"), indent=1)
