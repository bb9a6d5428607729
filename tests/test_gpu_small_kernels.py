"""Device parity of the small kernels the round-1 review found untested on the device:

  a2   Ellipsoid.__init__ from a covariance (dh_ell_from_cov) vs the reference's golden
       `be/*` ellipsoids, and its ValueError exit
  a4   improve_covar_mat (dh_improve_covar_mat) vs the reference's golden `icm/*` matrices
       (tests/test_ellipsoid.py:242-255 test_bounds)
  a11  scale_to_logvol: isotropic (golden `sc/*`), anisotropic (golden `aniso/*`), iterable
       targets, and the batched device form dh_enlarge_batch_dev
  a18  monte_carlo_logvol / unitcube_overlap vs the oracle on the same generator
"""
import math

import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib, backend
    c = _lib.Context(0)
    backend.set_backend(c)
    yield c
    backend.set_backend(None)


def canon_sign(axes):
    out = axes.copy()
    for k in range(axes.shape[1]):
        i = np.argmax(np.abs(out[:, k]))
        if out[i, k] < 0:
            out[:, k] = -out[:, k]
    return out


# ---- a2 ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_ell_from_cov_golden(ctx, name, golden_bounding):
    g = golden_bounding
    cov = g[f"{name}/be/cov"]
    loose = name == "flat10"  # condition number 1e11 (see test_gpu_rebuild)
    axes, axlens, ams, lvs = ctx.ell_from_cov(cov)
    np.testing.assert_allclose(lvs[0], float(g[f"{name}/be/logvol"]), rtol=0,
                               atol=1e-4 if loose else 1e-9)
    np.testing.assert_allclose(np.sort(axlens[0]), np.sort(g[f"{name}/be/axlens"]),
                               rtol=1e-5 if loose else 1e-9)
    np.testing.assert_allclose(ams[0], g[f"{name}/be/am"], rtol=0,
                               atol=(1e-3 if loose else 1e-8) * np.abs(g[f"{name}/be/am"]).max())
    ax = axes[0]
    np.testing.assert_allclose(ax @ ax.T, cov, rtol=0, atol=1e-10 * np.abs(cov).max())
    np.testing.assert_array_equal(ax, canon_sign(ax))
    assert np.all(np.diff(axlens[0]) >= 0)
    # the class constructor takes the same path (bounding.py:201-240)
    from dynesty_amd.bounding import HipEllipsoid
    e = HipEllipsoid(cov.shape[0], ctr=g[f"{name}/be/ctr"], cov=cov)
    np.testing.assert_allclose(e.logvol, float(g[f"{name}/be/logvol"]), rtol=0,
                               atol=1e-4 if loose else 1e-9)


def test_ell_from_cov_valueerror(ctx):
    """bounding.py:213-216: eigenvalues must be positive and finite."""
    from dynesty_amd.bounding import HipEllipsoid
    bad = np.diag([1.0, -0.5, 2.0])
    with pytest.raises(ValueError):
        ctx.ell_from_cov(bad)
    with pytest.raises(ValueError):
        HipEllipsoid(3, ctr=np.zeros(3), cov=bad)
    with pytest.raises(ValueError):
        ctx.ell_from_cov(np.zeros((4, 4)))
    with pytest.raises(ValueError):
        ctx.ell_from_cov(np.array([[1.0, np.nan], [np.nan, 1.0]]))
    # a stack: one bad matrix fails the call
    with pytest.raises(ValueError):
        ctx.ell_from_cov(np.stack([np.eye(3), bad]))


def test_ell_from_cov_vs_oracle_random(ctx):
    rng = np.random.default_rng(12)
    for d in (1, 2, 7, 25, 32, 44):
        a = rng.standard_normal((d, d + 3))
        cov = a @ a.T / (d + 3) * 0.01
        ref = B.make_ell(np.zeros(d), cov)
        axes, axlens, ams, lvs = ctx.ell_from_cov(cov)
        np.testing.assert_allclose(lvs[0], ref.logvol, rtol=0, atol=1e-9)
        np.testing.assert_allclose(axlens[0], ref.axlens, rtol=1e-9)
        np.testing.assert_allclose(ams[0], ref.am, rtol=0, atol=1e-8 * np.abs(ref.am).max())


# ---- a4 ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["zero", "rank1", "neg"])
def test_improve_covar_mat_golden(ctx, tag, golden_bounding):
    g = golden_bounding
    from dynesty_amd import bounding
    good, cov, am, axes = bounding.improve_covar_mat(g[f"icm/{tag}/in"])
    assert good == bool(g[f"icm/{tag}/good"])
    ref_cov, ref_am = g[f"icm/{tag}/cov"], g[f"icm/{tag}/am"]
    # rank1: one eigenvalue 30, the rest floored at 10 * 30 / 1e12: cov to 1e-9 of its
    # norm; am (eigenvalues up to 1/3e-10) relative to ITS norm
    np.testing.assert_allclose(cov, ref_cov, rtol=0, atol=1e-9 * max(np.abs(ref_cov).max(), 1e-300))
    # rank1: the floored eigenvalues (3e-10 next to 30) are only defined to eps * 30 / 3e-10 = 2e-5
    # relative by ANY eigensolver, and am = V diag(1/lam) V^T inherits that
    np.testing.assert_allclose(am, ref_am, rtol=0, atol=(2e-4 if tag == "rank1" else 1e-9) * np.abs(ref_am).max())
    np.testing.assert_allclose(axes @ axes.T, cov, rtol=0, atol=1e-9 * np.abs(cov).max())
    lam = np.linalg.eigvalsh(cov)
    assert lam.min() > 0 and lam.max() / lam.min() <= 1e12 * (1 + 1e-6)  # the reference's own test_bounds
    with pytest.raises(ValueError):
        bounding.improve_covar_mat(np.eye(3), ntries=5)


def test_improve_covar_mat_vs_oracle(ctx):
    rng = np.random.default_rng(3)
    mats = []
    for d in (3, 10, 25):
        a = rng.standard_normal((d, d))
        mats.append(a @ a.T)                       # well conditioned: untouched
        lowrank = a[:, :2] @ a[:, :2].T
        mats.append(lowrank)                       # rank 2: eigenvalue floor
    for m in mats:
        good_r, cov_r, am_r, axes_r = B.regularize_cov(m)
        good, cov, am, axes = ctx.improve_covar_mat(m)
        assert bool(good[0]) == good_r
        np.testing.assert_allclose(cov[0], cov_r, rtol=0, atol=1e-9 * np.abs(cov_r).max())
        np.testing.assert_allclose(am[0], am_r, rtol=0, atol=2e-4 * np.abs(am_r).max())


# ---- a11 --------------------------------------------------------------------------------
@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_scale_to_logvol_isotropic_golden(ctx, name, golden_bounding):
    """The reference's state after update (`mu/*`) -> scale_to_logvol(logvol + ln 1.25) -> `sc/*`."""
    g = golden_bounding
    from dynesty_amd.bounding import HipMultiEllipsoid
    m = HipMultiEllipsoid.__new__(HipMultiEllipsoid)
    m._set_arrays(g[f"{name}/mu/ctrs"], g[f"{name}/mu/covs"], g[f"{name}/mu/ams"],
                  g[f"{name}/mu/axes"], g[f"{name}/mu/axlens"], g[f"{name}/mu/logvol_ells"])
    m.ndim = m.ctrs.shape[1]
    m.logvol = float(g[f"{name}/mu/logvol"])
    m.scale_to_logvol(m.logvol + np.log(1.25))
    np.testing.assert_allclose(m.logvol, float(g[f"{name}/sc/logvol"]), rtol=0, atol=1e-11)
    np.testing.assert_allclose(m.logvol_ells, g[f"{name}/sc/logvol_ells"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(m.covs, g[f"{name}/sc/covs"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(m.ams, g[f"{name}/sc/ams"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(m.axes_ells, g[f"{name}/sc/axes"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(m.axlens_ells, g[f"{name}/sc/axlens"], rtol=1e-12)
    np.testing.assert_array_equal(m.ctrs, g[f"{name}/sc/ctrs"])


@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_scale_to_logvol_anisotropic_golden(ctx, name, golden_bounding):
    """bounding.py:257-275: a target volume that pushes axes against the sqrt(D)/2 cap.
    Start from the reference's bounding ellipsoid (`be/*`), compare with `aniso/*`."""
    g = golden_bounding
    from dynesty_amd.bounding import HipEllipsoid
    d = g[f"{name}/be/ctr"].shape[0]
    e = HipEllipsoid(d, ctr=g[f"{name}/be/ctr"], cov=g[f"{name}/be/cov"], am=g[f"{name}/be/am"],
                     axes=g[f"{name}/be/axes"], axlens=g[f"{name}/be/axlens"],
                     logvol=float(g[f"{name}/be/logvol"]))
    target = float(g[f"{name}/aniso/target"])
    e.scale_to_logvol(target)
    loose = name == "flat10"
    assert e.logvol == target
    np.testing.assert_allclose(e.axlens, g[f"{name}/aniso/axlens"], rtol=1e-10)
    np.testing.assert_allclose(e.axes, g[f"{name}/aniso/axes"], rtol=0,
                               atol=1e-10 * np.abs(g[f"{name}/aniso/axes"]).max())
    rc = g[f"{name}/aniso/cov"]
    np.testing.assert_allclose(e.cov, rc, rtol=0, atol=(1e-5 if loose else 1e-9) * np.abs(rc).max())
    ra = g[f"{name}/aniso/am"]
    np.testing.assert_allclose(e.am, ra, rtol=0, atol=(1e-3 if loose else 1e-8) * np.abs(ra).max())
    # no axis is pushed beyond sqrt(D)/2 (one that already exceeds it is left alone); the volume
    # is the target
    grown = e.axlens > g[f"{name}/be/axlens"] * (1 + 1e-12)
    assert np.all(e.axlens[grown] <= math.sqrt(d) / 2 * (1 + 1e-12))
    # (two5: an axis already exceeds the cap and the "target" is below the current volume -- the
    # reference then changes nothing but the stored logvol; the device follows, see the asserts above)
    from dynesty_amd.bounding import logvol_prefactor
    if target > float(g[f"{name}/be/logvol"]) and not np.any(g[f"{name}/be/axlens"] > math.sqrt(d) / 2):
        np.testing.assert_allclose(logvol_prefactor(d) + np.log(e.axlens).sum(), target, atol=1e-9)


def test_scale_to_logvol_iterable_targets(ctx, golden_bounding):
    """MultiEllipsoid.scale_to_logvol with one target per ellipsoid (bounding.py:485-487; the
    bootstrap expansion passes logvol_ells + ndim ln(expand))."""
    g = golden_bounding
    from dynesty_amd.bounding import HipMultiEllipsoid
    name = "c3"
    m = HipMultiEllipsoid.__new__(HipMultiEllipsoid)
    m._set_arrays(g[f"{name}/mu/ctrs"], g[f"{name}/mu/covs"], g[f"{name}/mu/ams"],
                  g[f"{name}/mu/axes"], g[f"{name}/mu/axlens"], g[f"{name}/mu/logvol_ells"])
    m.ndim = 2
    m.logvol = float(g[f"{name}/mu/logvol"])
    rng = np.random.default_rng(4)
    targets = m.logvol_ells + rng.uniform(0.0, 0.7, size=m.nells)
    ref = B.stack_ells([B.Ell(m.ctrs[i].copy(), m.covs[i].copy(), m.ams[i].copy(),
                              m.axes_ells[i].copy(), m.axlens_ells[i].copy(), float(m.logvol_ells[i]))
                        for i in range(m.nells)])
    ref = B.scale_multi_to_logvol(ref, targets)
    m.scale_to_logvol(targets)
    np.testing.assert_allclose(m.logvol_ells, targets, rtol=0, atol=0)
    np.testing.assert_allclose(m.logvol, ref.logvol, rtol=0, atol=1e-12)
    np.testing.assert_allclose(m.covs, ref.covs, rtol=1e-12)
    np.testing.assert_allclose(m.ams, ref.ams, rtol=1e-12)


@pytest.mark.parametrize("name", ["c3", "two5", "c2"])
def test_enlarge_batch_dev_golden(ctx, name, golden_bounding):
    """dh_enlarge_batch_dev (the form inside bench.py's timed region and the device NS loop): the
    reference's `mu/*` state laid out as runs x max_ells slots -> `sc/*`; slots past nells and the
    inactive copy must stay untouched."""
    g = golden_bounding
    ctrs = g[f"{name}/mu/ctrs"]
    m, d = ctrs.shape
    me, runs = m + 2, 3
    def slots(a):
        out = np.full((runs, me) + a.shape[1:], 7.25)
        out[:, :m] = a
        return out
    covs, ams, axes = slots(g[f"{name}/mu/covs"]), slots(g[f"{name}/mu/ams"]), slots(g[f"{name}/mu/axes"])
    axl, lv = slots(g[f"{name}/mu/axlens"]), slots(g[f"{name}/mu/logvol_ells"])
    nells = np.array([m, m, 0], dtype=np.int32)  # run 2 holds no bound yet
    d_n = ctx.to_device(nells)
    dev = [ctx.to_device(x) for x in (covs, ams, axes, axl, lv)]
    ctx._check(ctx.lib.dh_enlarge_batch_dev(ctx.handle, runs, me, d_n, d, *dev, math.log(1.25)))
    ctx.sync()
    out = [ctx.from_device(p, x.shape, np.float64) for p, x in zip(dev, (covs, ams, axes, axl, lv))]
    for r in (0, 1):
        np.testing.assert_allclose(out[0][r, :m], g[f"{name}/sc/covs"], rtol=1e-12)
        np.testing.assert_allclose(out[1][r, :m], g[f"{name}/sc/ams"], rtol=1e-12)
        np.testing.assert_allclose(out[2][r, :m], g[f"{name}/sc/axes"], rtol=1e-12)
        np.testing.assert_allclose(out[3][r, :m], g[f"{name}/sc/axlens"], rtol=1e-12)
        np.testing.assert_allclose(out[4][r, :m], g[f"{name}/sc/logvol_ells"], rtol=0, atol=1e-11)
    for o, x in zip(out, (covs, ams, axes, axl, lv)):
        np.testing.assert_array_equal(o[:, m:], x[:, m:])
        np.testing.assert_array_equal(o[2], x[2])
    for p in dev + [d_n]:
        ctx.free(p)


# ---- a18 --------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c3", "two5", "ring2"])
def test_monte_carlo_logvol_vs_oracle_same_generator(ctx, name):
    """bounding.py:608-630: the estimate is a function of the draws; on the same generator the
    device's draws are the oracle's (test_bound_draw_golden), so the estimates agree to rounding and
    the generator ends in the same state."""
    from dynesty_amd.bounding import HipMultiEllipsoid
    pts = inputs.cloud(name)
    d = pts.shape[1]
    m = HipMultiEllipsoid(d)
    m.update(pts)
    ref = B.stack_ells([B.Ell(m.ctrs[i].copy(), m.covs[i].copy(), m.ams[i].copy(),
                              m.axes_ells[i].copy(), m.axlens_ells[i].copy(), float(m.logvol_ells[i]))
                        for i in range(m.nells)])
    ndraws = 3000
    rs_dev, rs_ref = np.random.default_rng(99), np.random.default_rng(99)
    lv, ov = m.monte_carlo_logvol(ndraws=ndraws, rstate=rs_dev, return_overlap=True)
    draws = [B.multi_sample(ref, rs_ref, return_q=True) for _ in range(ndraws)]
    qsum = sum(1.0 / q for (_, _, q) in draws)
    lv_ref = np.log(qsum / ndraws) + ref.logvol
    ov_ref = sum(1.0 / q * (x.min() > 0 and x.max() < 1) for (x, _, q) in draws) / qsum
    np.testing.assert_allclose(lv, lv_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ov, ov_ref, rtol=0, atol=1e-12)
    assert rs_dev.bit_generator.state == rs_ref.bit_generator.state
    assert lv <= m.logvol + 1e-12
    # logvol-only form
    lv2 = m.monte_carlo_logvol(ndraws=500, rstate=np.random.default_rng(5), return_overlap=False)
    assert np.isfinite(lv2)


def test_unitcube_overlap_vs_oracle_same_generator(ctx):
    """bounding.py:336-343 for an ellipsoid sticking out of the cube."""
    from dynesty_amd.bounding import HipEllipsoid
    rng = np.random.default_rng(8)
    pts = np.array([0.9, 0.5, 0.1]) + 0.15 * rng.standard_normal((400, 3))
    e = HipEllipsoid(3)
    e.update(pts)
    ref = B.Ell(e.ctr.copy(), e.cov.copy(), e.am.copy(), e.axes.copy(), e.axlens.copy(), e.logvol)
    rs_dev, rs_ref = np.random.default_rng(17), np.random.default_rng(17)
    frac = e.unitcube_overlap(ndraws=2000, rstate=rs_dev)
    xs = np.array([B.ell_sample(ref, rs_ref) for _ in range(2000)])
    want = np.mean((xs.min(axis=1) > 0) & (xs.max(axis=1) < 1))
    assert 0.05 < want < 0.95
    np.testing.assert_allclose(frac, want, rtol=0, atol=1e-15)
    assert rs_dev.bit_generator.state == rs_ref.bit_generator.state
