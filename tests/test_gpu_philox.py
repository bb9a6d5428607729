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


# ---------------------------------------------------------------------------------------------------------
# Round 3: the throughput RNG mode of the slice samplers, of UniformBoundSampler / UnitCubeSampler and of the
# wave-per-walker kernels (dh_slice_batch_philox, dh_unif_batch_philox; ndim > 32 inside the same entry points).
# The reference's tests of these samplers (tests/test_sampling.py:24-100: marginal densities of slice chains
# inside a known region; tests/test_ellipsoid.py:14-89: radius^ndim ~ U(0, 1) Kolmogorov-Smirnov tests, the
# half-split of two overlapping ellipsoids) applied to device output.
# ---------------------------------------------------------------------------------------------------------
def ball_problem(ndim, radius=2.0, halfwidth=10.0):
    """iid Normal likelihood under a wide uniform prior: {logl > loglstar} is the ball |v| < radius, i.e.
    |u - 0.5| < radius / (2 halfwidth) -- a region whose uniform distribution has radius^ndim ~ U(0, 1)."""
    from dynesty_amd import problems
    prob = problems.gauss_iid(ndim, halfwidth, f"ball{ndim}")
    loglstar = float(prob.like_par[0] - 0.5 * radius * radius)
    return prob, loglstar, radius / (2.0 * halfwidth)


@pytest.mark.parametrize("ndim,principal,slices,k", [(2, False, 12, 40000), (5, False, 25, 30000), (5, True, 6, 30000),
                                                     (64, False, 150, 1500), (40, True, 20, 1500)])
def test_slice_chains_sample_the_contour_uniformly(ctx, ndim, principal, slices, k):
    """Slice chains (internal_samplers.py:593-855, 1075-1206) started at the centre of a ball-shaped likelihood
    contour must end uniformly distributed inside it: radius^ndim ~ U(0, 1) (the statistic of the reference's
    test_samples_single), every coordinate symmetric about the centre, nobody outside.  ndim = 64 / 40 run on the
    wave-per-walker kernels."""
    prob, loglstar, ru = ball_problem(ndim)
    u0 = np.full((k, ndim), 0.5)
    axes = np.eye(ndim) * ru
    out = ctx.slice_batch_philox(prob, u0, axes, 1.0, loglstar, slices, seed=99, sequence0=3, offset=0,
                                 principal=principal)
    assert np.all(out["logl"] > loglstar)
    r = np.linalg.norm(out["u"] - 0.5, axis=1) / ru
    assert r.max() < 1.0
    pval = scipy.stats.kstest(r**ndim, scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
    assert PVAL < pval < 1 - PVAL, pval
    for j in range(min(ndim, 8)):
        npos = (out["u"][:, j] > 0.5).sum()
        assert abs(npos - 0.5 * k) < 5 * np.sqrt(0.25 * k)
    # v and logl belong to the returned u
    v, ll = ctx.problem_eval(prob, out["u"][:512])
    np.testing.assert_allclose(out["logl"][:512], ll, rtol=1e-12, atol=1e-12)
    # work per chain: the same as the parity mode's on the same start points (call counts within 3 %)
    st = ctx.seed_children([5, ndim], 0, min(k, 4000))
    par = ctx.slice_batch(prob, u0[:len(st)], axes, 1.0, loglstar, slices, st, principal=principal)
    assert abs(out["ncalls"][:len(st)].mean() / par["ncalls"].mean() - 1.0) < 0.03


def test_slice_philox_keyed_reproducibility_and_doubling(ctx):
    prob, loglstar, ru = ball_problem(3)
    k = 3001
    u0 = np.full((k, 3), 0.5) + np.random.default_rng(1).uniform(-0.3, 0.3, size=(k, 3)) * ru
    kw = dict(seed=4242, offset=1 << 24)
    axes = np.eye(3) * ru * 0.05  # small steps: the stepping-out / doubling loops run long
    a = ctx.slice_batch_philox(prob, u0, axes, 1.0, loglstar, 5, sequence0=10, **kw)
    b = ctx.slice_batch_philox(prob, u0, axes, 1.0, loglstar, 5, sequence0=10, **kw)
    h0 = ctx.slice_batch_philox(prob, u0[:1500], axes, 1.0, loglstar, 5, sequence0=10, **kw)
    h1 = ctx.slice_batch_philox(prob, u0[1500:], axes, 1.0, loglstar, 5, sequence0=10 + 1500, **kw)
    for key in ("u", "logl", "ncalls", "n_expand", "n_contract"):
        np.testing.assert_array_equal(a[key], b[key])
        np.testing.assert_array_equal(a[key], np.concatenate([h0[key], h1[key]]))
    c = ctx.slice_batch_philox(prob, u0, axes, 1.0, loglstar, 5, sequence0=10, seed=4242, offset=2 << 24)
    assert np.abs(c["u"] - a["u"]).max() > 1e-3
    d = ctx.slice_batch_philox(prob, u0, axes, 1.0, loglstar, 5, sequence0=10, doubling=True, **kw)
    assert np.all(d["logl"] > loglstar)
    r = np.linalg.norm(d["u"] - 0.5, axis=1) / ru
    assert r.max() < 1.0 and d["n_expand"].mean() > 1.0


@pytest.mark.parametrize("ndim", [2, 10, 25, 48])
def test_unif_philox_single_ellipsoid_is_uniform(ctx, ndim):
    """UniformBoundSampler inside one ellipsoid with the threshold at -inf: the first candidate inside the
    cube is returned, so the points are Ellipsoid.sample draws (bounding.py:307-319): radius^ndim ~ U(0, 1)
    in the frame (test_samples_single), direction moments of a uniform ball.  ndim = 48: wide path."""
    from dynesty_amd import problems
    prob = problems.gauss_iid(ndim, 10.0, f"g{ndim}")
    rng = np.random.default_rng(ndim)
    k = 40000 if ndim <= 25 else 3000
    q, _ = np.linalg.qr(rng.standard_normal((ndim, ndim)))
    axes = q * rng.uniform(0.5, 2.0, size=ndim) * 0.01
    ctr = np.full(ndim, 0.5)
    out = ctx.unif_batch_philox(prob, -1e300, k, seed=7, sequence0=0, offset=0, ctrs=ctr[None], axes=axes[None])
    assert np.all(out["ncalls"] == 1)
    dr = np.linalg.solve(axes, (out["u"] - ctr).T).T
    r = np.linalg.norm(dr, axis=1)
    assert r.max() <= 1.0 + 1e-9
    pval = scipy.stats.kstest(r**ndim, scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
    assert PVAL < pval < 1 - PVAL, pval
    se = np.sqrt(1.0 / (ndim + 2) / k)
    assert np.abs(dr.mean(axis=0)).max() < 5 * se


@pytest.mark.parametrize("ndim", [2, 10])
def test_unif_philox_two_overlapping_ellipsoids(ctx, ndim):
    """The reference's test_sample (tests/test_ellipsoid.py:14-60): two unit balls 0.75 apart; draws from
    MultiEllipsoid.sample with the 1/q rejection are uniform in the union -- inside each ball radius^ndim is
    uniform and the plane between the centres splits the sample in halves."""
    from dynesty_amd import problems
    prob = problems.gauss_iid(ndim, 10.0, f"g{ndim}")
    k = 100000
    rad, shift = 0.02, 0.015
    c1 = np.full(ndim, 0.5)
    c2 = c1.copy()
    c2[0] += shift
    axes = np.stack([np.eye(ndim) * rad] * 2)
    ams = np.stack([np.eye(ndim) / rad**2] * 2)
    lv = np.zeros(2)
    out = ctx.unif_batch_philox(prob, -1e300, k, seed=11, ctrs=np.stack([c1, c2]), axes=axes, ams=ams, logvol_ells=lv)
    R = out["u"]
    d1 = np.linalg.norm(R - c1, axis=1) / rad
    d2 = np.linalg.norm(R - c2, axis=1) / rad
    assert np.all((d1 < 1) | (d2 < 1))
    for dist in (d1, d2):
        x = dist[dist < 1]**ndim
        pval = scipy.stats.kstest(x, scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
        assert PVAL < pval < 1 - PVAL, pval
    nhalf = (R[:, 0] > 0.5 + shift / 2.0).sum()
    assert abs(nhalf - 0.5 * k) < 5 * np.sqrt(0.5 * k)


@pytest.mark.parametrize("pname,ndim", [("C2", 25), ("C3", 2), ("W", 70)])
def test_unit_cube_philox(ctx, pname, ndim):
    """UnitCubeSampler (internal_samplers.py:364-441): uniform points of the cube above the threshold.  With
    the threshold at -inf every coordinate is U(0, 1) (KS per coordinate); with a real threshold all returned
    points lie above it and the call counts follow the geometric law of the accepted fraction."""
    from dynesty_amd import problems
    prob = inputs.problem(pname) if pname != "W" else problems.gauss_iid(ndim, 10.0, "g70")
    k = 20000 if ndim <= 32 else 2000
    out = ctx.unif_batch_philox(prob, -1e300, k, seed=21, sequence0=5)
    assert np.all(out["ncalls"] == 1) and out["u"].min() >= 0.0 and out["u"].max() < 1.0
    for j in range(min(ndim, 6)):
        pval = scipy.stats.kstest(out["u"][:, j], scipy.stats.uniform(loc=0.0, scale=1).cdf)[1]
        assert PVAL < pval < 1 - PVAL, (j, pval)
    c = np.corrcoef(out["u"][:-1, 0], out["u"][1:, 0])[0, 1]
    assert abs(c) < 5 / np.sqrt(k)
    if ndim == 2:
        thr = float(np.quantile(out["logl"], 0.7))
        o2 = ctx.unif_batch_philox(prob, thr, k, seed=22)
        assert np.all(o2["logl"] > thr)
        assert abs(o2["ncalls"].mean() - 1 / 0.3) < 5 * np.sqrt(0.7 / 0.09 / k) + 0.05
    a = ctx.unif_batch_philox(prob, -1e300, 1000, seed=21, sequence0=5)
    np.testing.assert_array_equal(a["u"], out["u"][:1000])


def test_philox_resident_loop_all_samplers(ctx):
    """dh_ns_ensemble with rng='philox' for rslice / slice and above 32 dimensions: the unit-cube phase and the
    proposals come from Philox streams; the evidence agrees with the analytic value like the parity mode's."""
    from dynesty_amd import problems
    for prob, kw, tol in ((inputs.problem("G5"), dict(bound="multi", sample="rslice", slices=5), 0.08),
                          (inputs.problem("G5"), dict(bound="multi", sample="slice", slices=3), 0.08)):
        r = ctx.ns_ensemble(prob, 16, 400, 64, entropy=[13], dlogz=0.05, rng="philox", **kw)
        assert np.all(r["status"] == 0)
        lz = r["logz"]
        se = lz.std(ddof=1) / np.sqrt(len(lz))
        assert abs(lz.mean() - prob.logz_truth) < 5 * se + tol, (prob, kw, lz.mean(), se)
        r2 = ctx.ns_ensemble(prob, 16, 400, 64, entropy=[13], dlogz=0.05, rng="philox", **kw)
        np.testing.assert_array_equal(r["logz"], r2["logz"])
    # above 32 dimensions (wave-per-walker kernels): 40-D with 400 live points is biased whatever the generator
    # (rwalk: too few live points for the walk; rslice in a single ellipsoid: the reference's own offset,
    # tests/golden/rslice_bias_ref.json), so the Philox mode is held to the parity mode
    prob = problems.gauss_iid(40, 10.0, "g40")
    for kw in (dict(bound="single", sample="rwalk", walks=60), dict(bound="single", sample="rslice", slices=43)):
        a = ctx.ns_ensemble(prob, 32, 400, 64, rng="philox", entropy=[14], dlogz=0.05, **kw)["logz"]
        b = ctx.ns_ensemble(prob, 32, 400, 64, rng="pcg64", entropy=[14], dlogz=0.05, **kw)["logz"]
        se = np.hypot(a.std(ddof=1), b.std(ddof=1)) / np.sqrt(32)
        assert abs(a.mean() - b.mean()) < 5 * se, (kw, a.mean(), b.mean(), se)
        assert 0.5 < a.std(ddof=1) / b.std(ddof=1) < 2.0


@pytest.mark.parametrize("pname", ["C1", "C3", "C2"])
def test_philox_unit_cube_four_lanes_per_walker_equals_one_lane(pname, monkeypatch):
    """Round 6: the throughput mode's unit-cube phase with four lanes per walker (try k of a walker =
    the 2 n words from offset + 2 n k of its keyed stream; DH_CUBE_FORM=2) against one walker per lane (=1): the same
    points, log-likelihoods and call counts for the same key, bit for bit -- also through whole resident runs in the
    Philox mode, whose unit-cube phase takes the four-lane form since this round (it ran one walker per lane, 0.74 ms
    per fill against 0.12, which made the "throughput" mode slower end to end than the parity mode)."""
    import inputs
    from dynesty_amd import _lib
    prob = inputs.problem(pname)
    ctxs = []
    for form in ("1", "2"):
        monkeypatch.setenv("DH_CUBE_FORM", form)
        ctxs.append(_lib.Context(0))
    k = 777
    u = np.random.default_rng(1).random((4000, prob.ndim))
    ll = prob.loglikelihood_many(prob.prior_transform_many(u))
    loglstar = float(np.quantile(ll, 0.97))
    a = ctxs[0].unif_batch_philox(prob, loglstar, k, seed=99, sequence0=1000, offset=8192)
    b = ctxs[1].unif_batch_philox(prob, loglstar, k, seed=99, sequence0=1000, offset=8192)
    for key in ("u", "v", "logl", "ncalls"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert a["ncalls"].max() > 40 and np.all(a["logl"] > loglstar)
    kw = dict(nlive=200, queue_size=48, walks=15, bound="single", entropy=[4, 4], dlogz=0.5, rng="philox")
    ra = ctxs[0].ns_ensemble(prob, 5, **kw)
    rb = ctxs[1].ns_ensemble(prob, 5, **kw)
    np.testing.assert_array_equal(ra["logz"], rb["logz"])
    np.testing.assert_array_equal(ra["ncall"], rb["ncall"])
