"""Seeded synthetic inputs shared by the golden-vector generator
(tools/make_golden.py) and the tests.  Everything is regenerated from the seed
(NumPy's PCG64 streams are stable across versions) so the committed fixtures
only need to hold *outputs*.
"""
import math

import numpy as np

from dynesty_amd import problems


def cloud(name):
    """Return an (n, d) float64 live-point-like cloud inside the unit cube."""
    if name == "c2":  # 25-D correlated cloud, N=2000 (rebuild input)
        rng = np.random.default_rng(101)
        d = 25
        cov = np.full((d, d), 0.4)
        np.fill_diagonal(cov, 1.0)
        chol = np.linalg.cholesky(cov)
        return 0.5 + 0.02 * (rng.standard_normal((2000, d)) @ chol.T)
    if name == "c3":  # eggbox-like: 13 modes in 2-D, N=5000
        rng = np.random.default_rng(102)
        even = [0.1, 0.5, 0.9]
        odd = [0.3, 0.7]
        ctrs = [(a, b) for a in even for b in even] + \
               [(a, b) for a in odd for b in odd]
        ctrs = np.array(ctrs)
        which = rng.integers(len(ctrs), size=5000)
        return ctrs[which] + 0.01 * rng.standard_normal((5000, 2))
    if name == "g3":  # 3-D blob, N=500 (C1-like)
        rng = np.random.default_rng(103)
        return 0.5 + 0.05 * rng.standard_normal((500, 3))
    if name == "two5":  # two separated blobs in 5-D, N=1000
        rng = np.random.default_rng(104)
        a = 0.3 + 0.02 * rng.standard_normal((600, 5))
        b = 0.7 + 0.03 * rng.standard_normal((400, 5))
        pts = np.vstack([a, b])
        return pts[rng.permutation(1000)]
    if name == "ring2":  # curved 2-D distribution -> many ellipsoids
        rng = np.random.default_rng(105)
        ang = rng.uniform(0, 2 * math.pi, size=3000)
        rad = 0.3 + 0.01 * rng.standard_normal(3000)
        return np.stack([0.5 + rad * np.cos(ang), 0.5 + rad * np.sin(ang)], 1)
    if name == "flat10":  # rank-3 data embedded in 10-D (improve_covar_mat path)
        rng = np.random.default_rng(106)
        basis = rng.standard_normal((3, 10))
        return 0.5 + 0.01 * rng.standard_normal((300, 3)) @ basis
    if name == "small4":  # fewer than 4*ndim points: no split attempted
        rng = np.random.default_rng(107)
        return rng.uniform(0.2, 0.8, size=(12, 4))
    if name == "egg13":  # c3 thinned: 13 well separated modes, N=1300
        return cloud("c3")[::4][:1300]
    if name == "c2s":  # c2 thinned to N=600 (25-D)
        return cloud("c2")[:600]
    if name == "g200":  # 200-D, N=4000 (C4 rebuild input)
        rng = np.random.default_rng(108)
        return 0.5 + 0.05 * rng.standard_normal((4000, 200))
    raise KeyError(name)


CLOUDS_SMALL = ["c2", "c3", "g3", "two5", "ring2", "flat10", "small4"]
# RadFriends / SupFriends fixtures: a blob, two separated blobs (clustering path), a ring
# (one chained cluster), the 13-mode eggbox-like cloud thinned to 1300 points, C2-like 25-D
CLOUDS_FRIENDS = ["g3", "two5", "ring2", "egg13", "c2s"]


def problem(name):
    if name in problems.BASELINE_PROBLEMS:
        return problems.BASELINE_PROBLEMS[name]()
    if name == "G5":
        return problems.gauss_corr(5, 0.7, 4.0, "G5")
    if name == "E3":
        return problems.eggbox(3, name="E3")
    if name == "N6":
        return problems.gauss_normal_prior(6, "N6")
    raise KeyError(name)


def walker_case(pname, nwalk, seed, shrink=0.5):
    """Starting points / loglstar / axes for ``nwalk`` walkers on ``pname``.

    Start points are drawn near the likelihood peak so a sizeable fraction of
    proposals is accepted; ``loglstar`` is a low quantile of their logl.
    Returns dict(u0 (k,d), loglstar, axes (d,d), scale).
    """
    prob = problem(pname)
    d = prob.ndim
    rng = np.random.default_rng(seed)
    if pname in ("C3", "E3"):
        u0 = 0.5 + 0.012 * rng.standard_normal((nwalk, d))
        spread = 0.012
    elif prob.prior_id == problems.PRIOR_NORMAL:
        u0 = 0.5 + 0.1 * rng.standard_normal((nwalk, d)) * shrink
        spread = 0.1 * shrink
    else:
        hw = prob.prior_par[0]
        u0 = 0.5 + (shrink / (2 * hw)) * rng.standard_normal((nwalk, d))
        spread = shrink / (2 * hw)
    u0 = np.clip(u0, 1e-3, 1 - 1e-3)
    logl = prob.loglikelihood_many(prob.prior_transform_many(u0))
    loglstar = float(np.quantile(logl, 0.05))
    # a random (non-diagonal) proposal frame of the right size
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    axes = (q * (spread * math.sqrt(d) * rng.uniform(0.8, 1.6, size=d)))
    keep = logl > loglstar
    return dict(u0=u0[keep], loglstar=loglstar, axes=axes, scale=0.7,
                problem=prob)


def blobs(d, sizes, sep, seed, sigma=0.001):
    rng = np.random.default_rng(seed)
    out = []
    for k, m in enumerate(sizes):
        c = np.full(d, 0.5)
        c[k % d] += sep * (1 if k % 2 == 0 else -1)
        c[(k + 3) % d] += 0.5 * sep * k
        A = rng.standard_normal((d, d)) * 0.4 * sigma
        out.append(c + rng.standard_normal((m, d)) * sigma + rng.standard_normal((m, d)) @ A)
    pts = np.vstack(out)
    return pts[rng.permutation(len(pts))]


def wide_walker_case(d, k, seed):
    """Start points, threshold and frame of the wide-D walker tests (iid Normal likelihood, Normal
    prior via ndtri): shared by tools/make_golden.py (the real reference) and the tests."""
    from dynesty_amd import problems
    prob = problems.gauss_normal_prior(d, "C4")
    rng = np.random.default_rng(seed)
    u0 = np.clip(0.5 + 0.08 * rng.standard_normal((k, d)), 0.02, 0.98)
    logl0 = prob.loglikelihood_many(prob.prior_transform_many(u0))
    loglstar = float(logl0.min() - 5.0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    axes = q * (0.08 * np.sqrt(d) * rng.uniform(0.8, 1.4, size=d))
    return dict(problem=prob, u0=u0, loglstar=loglstar, axes=axes)
