"""Shape sweep of the device-resident loop: every sampler / bound / RNG mode / rebuild period at odd sizes; checks status,
ln Z against the analytic value and (PCG64) the invariance of every run's record under the rebuild period.
usage: fuzz_loop.py [seed]"""
import itertools, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynesty_amd import _lib, problems
ctx = _lib.Context(0)
rng0 = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
cases = []
for d in (2, 3, 5, 9, 13, 25, 32, 40):
    for sample in ("rwalk", "rslice", "slice", "unif"):
        if sample == "unif" and d > 13:
            continue
        if sample == "slice" and d > 13:
            continue
        for bound in ("single", "multi"):
            cases.append((d, sample, bound))
rng0.shuffle(cases)
t0 = time.time()
for d, sample, bound in cases[:int(os.environ.get("NCASE", "40"))]:
    prob = problems.gauss_iid(d, 6.0, f"g{d}") if rng0.random() < 0.5 or d < 3 else problems.gauss_corr(d, 0.3, 5.0, f"c{d}")
    nlive = int(rng0.choice([60, 150, 333, 700]))
    if bound == "multi":
        nlive = max(nlive, 8 * d)
    K = int(rng0.choice([1, 7, 48, 64, 100, 257]))
    runs = int(rng0.choice([1, 3, 8]))
    mode = str(rng0.choice(["pcg64", "philox"]))
    kw = dict(bound=bound, sample=sample, dlogz=0.5, entropy=[int(rng0.integers(1 << 30))], rng=mode)
    if sample == "rwalk":
        kw["walks"] = d + 20  # the reference's default
    elif sample in ("rslice", "slice"):
        kw["slices"] = 3 + d if sample == "rslice" else 3  # the reference's defaults
    # round 4: the forced update inside the fill, boundary flags (one coordinate stays hard, as the reference needs),
    # a non-default update interval, a live set beyond the LDS-resident consumption
    if sample != "unif" and rng0.random() < 0.5:
        kw["forced_exact"] = True
    if sample in ("rwalk", "unif") and d >= 3 and rng0.random() < 0.4:
        kw["periodic"], kw["reflective"] = [0], [d - 1]
    if rng0.random() < 0.3:
        kw["update_interval"] = float(rng0.choice([0.6, 1.7]))
    if d <= 5 and sample == "rwalk" and rng0.random() < 0.5:
        nlive = int(rng0.choice([8500, 12000]))
        K = int(rng0.choice([64, 257]))
    tag = dict(d=d, sample=sample, bound=bound, nlive=nlive, K=K, runs=runs, rng=mode, prob=prob.name,
               opts={k: v for k, v in kw.items() if k in ("forced_exact", "periodic", "update_interval")})
    try:
        a = ctx.ns_ensemble(prob, runs, nlive, K, rebuild_every=1, **kw)
        b = ctx.ns_ensemble(prob, runs, nlive, K, rebuild_every=0, **kw)
        ok = bool((a["status"] == 0).all() and (b["status"] == 0).all())
        same = True
        if mode == "pcg64":
            same = bool(np.array_equal(a["logz"], b["logz"]) and np.array_equal(a["ncall"], b["ncall"]))
            ok = ok and same
        # generous: a run's own error estimate (plus the dlogz truncation and the samplers' known biases at few live points)
        z = np.abs(b["logz"] - prob.logz_truth) / np.maximum(b["logzerr"], 0.05)
        ok = ok and bool(np.all(z < 8) and np.all(np.isfinite(b["logz"])))
        if not ok:
            bad += 1
        print(json.dumps(dict(tag, ok=ok, same=same, logz=round(float(b["logz"].mean()), 3), truth=round(float(prob.logz_truth), 3),
                              zmax=round(float(z.max()), 2), nfills=int(b["nfills"]))), flush=True)
    except Exception as exc:
        bad += 1
        print(json.dumps(dict(tag, ok=False, error=repr(exc)[:200])), flush=True)
print(json.dumps(dict(cases=len(cases), bad=bad, secs=round(time.time() - t0, 1))))
