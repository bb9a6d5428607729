#!/usr/bin/env python
"""Fold the JSON lines written by tools/ref_c4_runs.py (real dynesty, BASELINE config C4) into
tests/golden/c4_logz_ref.json: usage ref_c4_collect.py DIR  (DIR holds c4_K<K>_*.jsonl); runs already in the
file are kept, duplicates (same K and seed) dropped, per-K ensemble statistics recomputed."""
import glob
import json
import os
import sys

import numpy as np

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "c4_logz_ref.json")
out = json.load(open(path))
seen = {(int(r["K"]), int(r["seed"])) for r in out["runs"]}
for f in sorted(glob.glob(os.path.join(src, "c4_K*_*.jsonl"))):
    for line in open(f):
        if not line.strip():
            continue
        r = json.loads(line)
        key = (int(r["K"]), int(r["seed"]))
        if key not in seen:
            seen.add(key)
            out["runs"].append(r)
out["runs"].sort(key=lambda r: (int(r["K"]), int(r["seed"])))
out["ensembles"] = {}
for K in sorted({int(r["K"]) for r in out["runs"]}):
    rows = [r for r in out["runs"] if int(r["K"]) == K]
    lz = np.array([r["logz"] for r in rows])
    out["ensembles"][f"K{K}"] = dict(
        K=K, n=len(rows), mean=float(lz.mean()), std=float(lz.std(ddof=1)) if len(rows) > 1 else None,
        se=float(lz.std(ddof=1) / np.sqrt(len(rows))) if len(rows) > 1 else None,
        mean_logzerr=float(np.mean([r["logzerr"] for r in rows])), mean_niter=float(np.mean([r["niter"] for r in rows])),
        mean_ncall=float(np.mean([r["ncall"] for r in rows])), mean_seconds_1core=float(np.mean([r["seconds"] for r in rows])))
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["ensembles"], indent=1))
