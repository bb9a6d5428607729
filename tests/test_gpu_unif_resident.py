"""The uniform sampler and the bootstrap expansion inside the device-resident loop (dh_ns_ensemble
sampler 6 / 7, dh_bootstrap_expand): the reference's default for low-D problems (dynesty.py:169-200:
sample='unif' -> enlarge 1, bootstrap 5; BASELINE config C1)."""
import numpy as np
import pytest

import inputs
from oracle import nested_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def _clouds(runs, n, d, seed, two=False):
    g = np.random.default_rng(seed)
    pts = 0.5 + 0.05 * g.standard_normal((runs, n, d)) @ np.diag(np.linspace(0.3, 1.0, d))
    if two:
        pts[:, ::2] += 0.25
        pts[:, 1::2] -= 0.2
    return np.ascontiguousarray(pts)


@pytest.mark.parametrize("n,d,multi,two", [(200, 3, False, False), (500, 3, True, True), (300, 10, False, False),
                                           (1000, 2, True, True), (64, 25, False, False),
                                           # above d = 44: the replicas go through the wide constructions one by one
                                           (300, 48, False, False), (420, 48, True, True)])
def test_bootstrap_expand_vs_oracle(ctx, n, d, multi, two):
    """Every replica's resampling mask is the NumPy draw of the oracle's generator (bit for bit: the expansion
    factor is the distance of one particular left-out point), its bound the oracle's construction."""
    runs, B = 4, 5
    pts = _clouds(runs, n, d, 100 + n + d, two)
    ent = np.random.default_rng(n * d).integers(0, 2**63, size=(runs, 4), dtype=np.uint64)
    got = ctx.bootstrap_expand(pts, ent, B, multi)
    want = np.array([nested_ref.boot_expand(pts[r], ent[r], B, multi) for r in range(runs)])
    assert np.all(want >= 1.0)
    assert np.any(want > 1.0)
    np.testing.assert_allclose(got, want, rtol=1e-9)


def test_bootstrap_expand_repairs(ctx):
    """_bootstrap_points' repair of a replica that drew every point (n_in > n - 1 -> the first is left out): with
    four points that is one replica in eleven."""
    runs = 32
    pts = np.array([[[0.2, 0.3], [0.7, 0.6], [0.4, 0.8], [0.5, 0.1]]] * runs)
    ent = np.arange(4 * runs, dtype=np.uint64).reshape(runs, 4) * np.uint64(0x9E3779B97F4A7C15)
    raw = np.array([[len(np.unique(nested_ref.boot_generator(ent[r], b).integers(4, size=4))) for b in range(2)]
                    for r in range(runs)])
    ok = (raw >= 2).all(axis=1)  # (a replica of ONE distinct point can leave nothing out: the reference then fails)
    assert ok.sum() > 20 and (raw[ok] == 4).any()
    got, nin = ctx.bootstrap_expand(pts[ok], ent[ok], 2, False, want_n_in=True)
    want = np.array([nested_ref.boot_expand(pts[r], ent[r], 2, False) for r in np.nonzero(ok)[0]])
    # (two- and three-point "clouds" in 2-D: regularised, near-singular covariances -- the distances of the
    # left-out points are of order 1e5 and good to 1e-5 only)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    np.testing.assert_array_equal(nin, np.minimum(raw[ok], 3))


@pytest.mark.parametrize("pname,bound,nlive,K,rng", [
    ("C1", "single", 500, 64, "pcg64"), ("C1", "multi", 500, 64, "pcg64"), ("C1", "single", 500, 100, "philox"),
    ("C3", "multi", 1000, 128, "pcg64"), ("G5", "multi", 400, 32, "philox")])
def test_unif_resident_logz(ctx, pname, bound, nlive, K, rng):
    """Reference defaults (bootstrap 5, enlarge 1).  K = 100: a queue that is not a multiple of the wavefront."""
    prob = inputs.problem(pname)
    r = ctx.ns_ensemble(prob, 16, nlive, K, bound=bound, sample="unif", entropy=[5, K], dlogz=0.05, rng=rng)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    se = lz.std(ddof=1) / np.sqrt(len(lz))
    assert abs(lz.mean() - prob.logz_truth) < 5 * se + 0.08, (lz.mean(), se)
    assert 0.4 < lz.std(ddof=1) / r["logzerr"].mean() < 2.5
    # a bound every nlive calls once the unit-cube phase is over (internal_samplers.py:88-94; most of a low-D
    # run's calls are spent in that phase)
    assert np.all(r["nbound"] >= 5)


def test_unif_resident_enlarge_instead_of_bootstrap(ctx):
    prob = inputs.problem("C1")
    kw = dict(bound="single", sample="unif", entropy=[9], dlogz=0.05)
    a = ctx.ns_ensemble(prob, 16, 500, 64, enlarge=1.25, **kw)   # -> bootstrap 0
    b = ctx.ns_ensemble(prob, 16, 500, 64, **kw)                 # -> bootstrap 5
    for r in (a, b):
        assert np.all(r["status"] == 0)
        lz = r["logz"]
        assert abs(lz.mean() - prob.logz_truth) < 5 * lz.std(ddof=1) / 4 + 0.08
    # different bounds, different runs
    assert not np.array_equal(a["ncall"], b["ncall"])
    with pytest.raises(ValueError):
        ctx.ns_ensemble(prob, 2, 500, 64, enlarge=1.25, bootstrap=5, **kw)


def test_unif_resident_deterministic_and_sharding_independent(ctx):
    prob = inputs.problem("C1")
    kw = dict(nlive=300, queue_size=48, bound="multi", sample="unif", entropy=[3, 1, 4], dlogz=0.5)
    a = ctx.ns_ensemble(prob, 6, **kw)
    b = ctx.ns_ensemble(prob, 6, **kw)
    np.testing.assert_array_equal(a["logz"], b["logz"])
    lo = ctx.ns_ensemble(prob, 3, first_run=0, **kw)
    hi = ctx.ns_ensemble(prob, 3, first_run=3, **kw)
    np.testing.assert_array_equal(np.concatenate([lo["logz"], hi["logz"]]), a["logz"])
    np.testing.assert_array_equal(np.concatenate([lo["ncall"], hi["ncall"]]), a["ncall"])
