"""Throughput RNG mode of the rwalk kernel (hiprand Philox4x32-10, dh_rwalk_batch_philox): not
stream-compatible with the reference, so it is validated statistically --

  * the reference's own KS tests of tests/test_ellipsoid.py (radius^ndim of a draw in the unit ball
    ~ U(0,1): test_sample / test_samples_single) applied to the kernel's one-step proposals,
    plus the direction moments of a uniform ball (mean 0, covariance I / (ndim + 2));
  * the 45-step chain against the PARITY mode on the same start points: acceptance fractions and the
    distribution of the final log-likelihoods (two-sample KS);
  * keyed reproducibility: same (seed, sequence0, offset) -> identical results; the walker key is
    sequence0 + index, so a batch equals its two halves.
"""
import numpy as np
import pytest
import scipy.stats

import inputs

pytestmark = pytest.mark.gpu
PVAL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


@pytest.mark.parametrize("ndim", [2, 10, 25])
def test_one_step_proposals_are_uniform_in_the_frame_ball(ctx, ndim):
    """walks = 1, threshold -inf: u' = u0 + scale * axes @ drhat with drhat uniform in the unit ball
    (bounding.py:1288-1297).  In the frame's coordinates |dr|^ndim ~ U(0,1) -- the statistic of the
    reference's test_samples_single."""
    from dynesty_amd import problems
    prob = problems.gauss_corr(ndim, 0.3, 50.0, "wide-prior")  # wide prior: nothing leaves the cube
    rng = np.random.default_rng(ndim)
    k = 100000
    q, _ = np.linalg.qr(rng.standard_normal((ndim, ndim)))
    axes = q * rng.uniform(0.5, 2.0, size=ndim) * 1e-3
    u0 = np.full((k, ndim), 0.5)
    out = ctx.rwalk_batch_philox(prob, u0, axes, 1.0, -1e300, 1, seed=12345, sequence0=7, offset=0)
    assert np.all(out["accept"] == 1)
    dr = np.linalg.solve(axes, (out["u"] - u0).T).T
    r = np.linalg.norm(dr, axis=1)
    assert r.max() <= 1.0 + 1e-9
    pval = scipy.stats.kstest(r**ndim, scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
    assert PVAL < pval < 1 - PVAL
    # direction: mean 0, covariance I / (ndim + 2), to 5 standard errors
    se = np.sqrt(1.0 / (ndim + 2) / k)
    assert np.abs(dr.mean(axis=0)).max() < 5 * se
    cov = dr.T @ dr / k
    target = np.eye(ndim) / (ndim + 2)
    assert np.abs(cov - target).max() < 6 * np.sqrt(2.0) / (ndim + 2) / np.sqrt(k) + 1e-4 / (ndim + 2)
    # symmetric proposal: the sign pattern of every coordinate is a fair coin
    for j in range(ndim):
        npos = (dr[:, j] > 0).sum()
        assert abs(npos - 0.5 * k) < 5 * np.sqrt(0.25 * k)
    # each walker's stream is its own: consecutive walkers are uncorrelated
    c = np.corrcoef(dr[:-1, 0], dr[1:, 0])[0, 1]
    assert abs(c) < 5 / np.sqrt(k)


def test_chain_statistics_match_the_parity_mode(ctx):
    """BASELINE C2 walkers, 45 steps: the Philox chain and the PCG64 (parity) chain are draws of the
    same Markov kernel."""
    case = inputs.walker_case("C2", 40000, 4711)
    prob = case["problem"]
    u0 = case["u0"][:30000]
    k = len(u0)
    states = ctx.seed_children([99, 1], 0, k)
    ref = ctx.rwalk_batch(prob, u0, case["axes"], case["scale"], case["loglstar"], 45, states)
    out = ctx.rwalk_batch_philox(prob, u0, case["axes"], case["scale"], case["loglstar"], 45, seed=5)
    assert np.all(out["accept"] + out["reject"] == 45)
    fa, fb = ref["accept"].mean() / 45, out["accept"].mean() / 45
    se = np.sqrt(2 * 0.25 / (45 * k))
    assert abs(fa - fb) < 6 * se + 2e-3, (fa, fb)
    assert np.all(out["logl"] > case["loglstar"])
    p = scipy.stats.ks_2samp(ref["logl"], out["logl"])[1]
    assert p > PVAL, p
    p = scipy.stats.ks_2samp(ref["accept"], out["accept"])[1]
    assert p > PVAL, p
    # v is the prior transform of u, logl its likelihood
    v, ll = ctx.problem_eval(prob, out["u"])
    np.testing.assert_allclose(out["v"], v, rtol=0, atol=1e-13)
    np.testing.assert_allclose(out["logl"], ll, rtol=1e-12)


def test_keyed_reproducibility(ctx):
    case = inputs.walker_case("G5", 3000, 5)
    prob, u0 = case["problem"], case["u0"][:2000]
    kw = dict(scale=case["scale"], loglstar=case["loglstar"], walks=20)
    a = ctx.rwalk_batch_philox(prob, u0, case["axes"], seed=77, sequence0=100, offset=3, **kw)
    b = ctx.rwalk_batch_philox(prob, u0, case["axes"], seed=77, sequence0=100, offset=3, **kw)
    for key in ("u", "logl", "accept"):
        np.testing.assert_array_equal(a[key], b[key])
    # walker key = sequence0 + index: a batch equals its halves
    h1 = ctx.rwalk_batch_philox(prob, u0[:1000], case["axes"], seed=77, sequence0=100, offset=3, **kw)
    h2 = ctx.rwalk_batch_philox(prob, u0[1000:], case["axes"], seed=77, sequence0=1100, offset=3, **kw)
    np.testing.assert_array_equal(np.concatenate([h1["u"], h2["u"]]), a["u"])
    # another offset / seed: different draws
    c = ctx.rwalk_batch_philox(prob, u0, case["axes"], seed=77, sequence0=100, offset=100000, **kw)
    d = ctx.rwalk_batch_philox(prob, u0, case["axes"], seed=78, sequence0=100, offset=3, **kw)
    assert (c["u"] != a["u"]).any(axis=1).mean() > 0.9
    assert (d["u"] != a["u"]).any(axis=1).mean() > 0.9


def test_boundaries_and_partial_clustering(ctx):
    """periodic / reflective coordinates and ncdim < ndim take the same code as the parity kernel."""
    from dynesty_amd import _lib
    prob = inputs.problem("G5")
    rng = np.random.default_rng(3)
    k = 20000
    u0 = rng.uniform(0.02, 0.98, size=(k, 5))
    axes = np.eye(3) * 0.3
    bc = np.array([_lib.BC_PERIODIC, _lib.BC_REFLECT, _lib.BC_HARD, _lib.BC_HARD, _lib.BC_HARD], dtype=np.int8)
    out = ctx.rwalk_batch_philox(prob, u0, axes, 1.0, -1e300, 5, seed=9, ncdim=3, bc=bc)
    assert np.all(out["accept"] + out["reject"] == 5)
    u = out["u"]
    assert u.min() > 0 and u.max() < 1
    # the non-clustered coordinates are redrawn uniformly at every step (internal_samplers.py:1007-1010)
    moved = out["accept"] > 0
    for j in (3, 4):
        p = scipy.stats.kstest(u[moved, j], scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
        assert p > PVAL
