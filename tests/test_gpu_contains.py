"""K5 membership kernel vs the oracle and the reference's golden index lists
(reference tests/test_ellipsoid.py:106-133 test_overlap is the model)."""
import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def unpack(mask, k):
    m = mask.shape[0]
    bits = np.unpackbits(mask.view(np.uint8).reshape(m, -1), axis=1,
                         bitorder="little")[:, :k]
    return bits.astype(bool)


@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_within_golden(ctx, name, golden_bounding):
    g = golden_bounding
    ctrs, ams = g[f"{name}/mu/ctrs"], g[f"{name}/mu/ams"]
    probes = g[f"{name}/kat/probes"]
    k = probes.shape[0]
    count, mask, quad = ctx.contains(probes, ctrs, ams, mode=0, want_mask=True,
                                     want_quad=True)
    inside = unpack(mask, k)  # (m, k)
    within = [np.nonzero(inside[:, i])[0] for i in range(k)]
    np.testing.assert_array_equal(np.concatenate(within),
                                  g[f"{name}/kat/within_flat"])
    np.testing.assert_array_equal(count, g[f"{name}/kat/within_count"])
    np.testing.assert_array_equal(count > 0, g[f"{name}/kat/contains"])
    skip0 = count - inside[0].astype(np.int32)
    np.testing.assert_array_equal(skip0, g[f"{name}/kat/overlap_skip0"])
    want_q = np.array([B.multi_quadforms(p, ctrs, ams) for p in probes])
    np.testing.assert_allclose(quad, want_q, rtol=1e-12, atol=1e-13)


def test_two_spheres_brute_force(ctx):
    """1e4 points against two unit spheres, exact index lists
    (reference tests/test_ellipsoid.py:106-133)."""
    rng = np.random.default_rng(3)
    for ndim in (2, 10):
        ctrs = np.zeros((2, ndim))
        ctrs[0, 0] = -0.7
        ctrs[1, 0] = 0.7
        ams = np.array([np.eye(ndim), np.eye(ndim)])
        x = rng.uniform(-2, 2, size=(10000, ndim))
        count, mask, _ = ctx.contains(x, ctrs, ams, mode=0, want_mask=True)
        inside = unpack(mask, x.shape[0])
        brute = np.stack([((x - c)**2).sum(1) < 1 for c in ctrs])
        np.testing.assert_array_equal(inside, brute)
        np.testing.assert_array_equal(count, brute.sum(0))


@pytest.mark.parametrize("name", ["g3", "c2"])
def test_single_mode(ctx, name, golden_bounding):
    """Ellipsoid.contains uses sqrt(q) <= 1 (bounding.py:302-305)."""
    g = golden_bounding
    e = B.bounding_ellipsoid(inputs.cloud(name))
    probes = g[f"{name}/kat/probes"]
    count, _, _ = ctx.contains(probes, e.ctr[None], e.am[None], mode=1)
    np.testing.assert_array_equal(count > 0, g[f"{name}/single/contains"])


def test_ragged_sizes(ctx):
    rng = np.random.default_rng(4)
    for k in (1, 63, 64, 65, 1000):
        d, m = 3, 5
        ctrs = rng.uniform(0.3, 0.7, size=(m, d))
        a = rng.standard_normal((m, d, d))
        ams = np.einsum('mij,mkj->mik', a, a) * 4 + np.eye(d)
        x = rng.uniform(0, 1, size=(k, d))
        count, mask, quad = ctx.contains(x, ctrs, ams, want_mask=True,
                                         want_quad=True)
        want = np.array([B.multi_quadforms(p, ctrs, ams) for p in x])
        np.testing.assert_allclose(quad, want, rtol=1e-12)
        np.testing.assert_array_equal(count, (want < 1).sum(1))
        np.testing.assert_array_equal(unpack(mask, k), (want < 1).T)


@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_class_contains_scalar_and_batch_golden(ctx, name, golden_bounding):
    """Bound.contains(x) for one point (host scalar query) and contains_many (membership kernel) both
    give the reference's golden verdicts (MultiEllipsoid.contains / Ellipsoid.contains on the probes)."""
    from dynesty_amd import backend
    from dynesty_amd.bounding import HipEllipsoid, HipMultiEllipsoid
    g = golden_bounding
    probes = g[f"{name}/kat/probes"]
    backend.set_backend(ctx)
    try:
        m = HipMultiEllipsoid.__new__(HipMultiEllipsoid)
        m._set_arrays(g[f"{name}/mu/ctrs"], g[f"{name}/mu/covs"], g[f"{name}/mu/ams"], g[f"{name}/mu/axes"],
                      g[f"{name}/mu/axlens"], g[f"{name}/mu/logvol_ells"])
        want = g[f"{name}/kat/contains"]
        np.testing.assert_array_equal([m.contains(p) for p in probes], want)
        np.testing.assert_array_equal(m.contains_many(probes), want)
        d = probes.shape[1]
        e = HipEllipsoid(d)
        e.update(inputs.cloud(name))
        got_one = np.array([e.contains(p) for p in probes])
        np.testing.assert_array_equal(got_one, e.contains_many(probes))
        np.testing.assert_array_equal(got_one, g[f"{name}/single/contains"])
    finally:
        backend.set_backend(None)


def test_membership_test_of_the_resident_loop_lds_form_equals_scalar_form():
    """Round 6: the start points' membership test of the resident loop (contains_runs_kernel) with the ellipsoid staged in
    LDS by the workgroup -- for queues of a multiple of 256 entries at D >= 9 -- against the form that fetches it
    through the scalar cache (DH_CONTAINS_LDS=0): the same multiply-adds in the same order, so whole runs are the same,
    iteration for iteration: every record field bit for bit (multi and single bounds, two dimensions)."""
    import os
    import inputs
    from dynesty_amd import _lib, problems
    c = _lib.Context(0)
    for prob, bound, K in ((problems.gauss_corr(13, 0.4, 5.0, "corr13"), "multi", 256),
                           (inputs.problem("C2"), "single", 512), (inputs.problem("C2"), "multi", 256)):
        outs = []
        for v in ("1", "0"):
            os.environ["DH_CONTAINS_LDS"] = v
            try:
                outs.append(c.ns_ensemble(prob, 6, 600, K, walks=25, bound=bound, entropy=[5], want_dead_logl=True))
            finally:
                del os.environ["DH_CONTAINS_LDS"]
        a, b = outs
        assert int(a["nbound"].sum()) > 6
        for k in ("logz", "logzerr", "niter", "ncall", "nbound"):
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)
        for r, n in enumerate(a["niter"]):  # (the store is only written up to the run's length)
            np.testing.assert_array_equal(a["dead_logl"][r][:n], b["dead_logl"][r][:n])
