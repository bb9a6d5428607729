#!/usr/bin/env python
"""logZ gate for BASELINE config C2 on the device: N seeds of the full static
run (25-D rho=0.4 Normal, nlive=2000, multi/rwalk walks=45, dlogz=0.01) through
the dynesty-free driver.  Reference (same settings, seed 21): -57.4541;
analytic truth: -57.5646 (SURVEY.md sections 6, 8c)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from dynesty_amd import nested, problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
rows = []
for s in range(n):
    t = time.perf_counter()
    r = nested.run_static(prob, nlive=2000, bound='multi', sample='rwalk',
                          walks=45, queue_size=K,
                          rstate=np.random.default_rng(1000 + s), dlogz=0.01)
    dt = time.perf_counter() - t
    rows.append(dict(seed=1000 + s, logz=r.logz, logzerr=r.logzerr,
                     niter=r.niter, ncall=r.ncall, nbound=r.nbound,
                     seconds=dt, calls_per_s=r.ncall / dt))
    print(json.dumps(rows[-1]))
lz = np.array([x["logz"] for x in rows])
print(json.dumps(dict(
    n=n, queue_size=K, mean_logz=lz.mean(), se=lz.std(ddof=1) / np.sqrt(n),
    std=lz.std(ddof=1), truth=-57.5646, reference_seed21=-57.4541,
    mean_minus_truth=lz.mean() + 57.5646,
    mean_minus_reference=lz.mean() + 57.4541,
    mean_calls_per_s=float(np.mean([x["calls_per_s"] for x in rows])),
    mean_seconds=float(np.mean([x["seconds"] for x in rows])))))
