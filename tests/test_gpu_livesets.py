"""Rebuild kernels on live sets captured from REAL reference runs (tests/golden/livesets.npz, tools/make_golden.py
livesets) -- uniform-in-contour shells, not Gaussian clouds -- against what the reference's
own MultiEllipsoid.update produced from them inside the run: C2 (2000 x 25) at bound updates 1 / 8 / 24 and C3
(eggbox, 5000 x 2, 13-14 ellipsoids) at updates 1 / 6 / 16."""
import os

import numpy as np
import pytest

from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "livesets.npz")


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


@pytest.mark.parametrize("tag,i", [("C2", 0), ("C2", 1), ("C2", 2), ("C3", 0), ("C3", 1), ("C3", 2)])
def test_rebuild_of_a_captured_live_set(ctx, tag, i):
    g = np.load(G)
    pts = g[f"{tag}/{i}/live_u"]
    got = ctx.rebuild(pts, multi=True, want_labels=True)
    m = int(g[f"{tag}/{i}/nells"])
    assert got["nells"] == m
    ref_c, ref_cov, ref_lv = g[f"{tag}/{i}/ctrs"], g[f"{tag}/{i}/covs"], g[f"{tag}/{i}/logvol_ells"]
    # the reference's list order depends on LAPACK's eigenvector signs: match by centre
    dist = np.linalg.norm(got["ctrs"][:, None, :] - ref_c[None, :, :], axis=2)
    p = dist.argmin(axis=1)
    assert sorted(p.tolist()) == list(range(m))
    for a in range(m):
        b = p[a]
        np.testing.assert_allclose(got["ctrs"][a], ref_c[b], rtol=0, atol=1e-13)
        np.testing.assert_allclose(got["covs"][a], ref_cov[b], rtol=1e-9, atol=1e-9 * np.abs(ref_cov[b]).max())
        np.testing.assert_allclose(got["logvol_ells"][a], ref_lv[b], rtol=0, atol=1e-9)
    from scipy.special import logsumexp
    np.testing.assert_allclose(logsumexp(got["logvol_ells"]), float(g[f"{tag}/{i}/logvol"]), rtol=0, atol=1e-9)
    # every live point strictly inside the union; the clusters are the oracle's leaves
    lab = got["labels"]
    for a in range(m):
        d = pts[lab == a] - got["ctrs"][a]
        assert (np.einsum('ij,jk,ik->i', d, got["ams"][a], d) < 1.0).all()
    # and in the device's own sign convention the ORDER is the oracle's too
    old = B.CANON_SIGNS
    try:
        B.CANON_SIGNS = True
        ref = B.multi_update(pts)
    finally:
        B.CANON_SIGNS = old
    np.testing.assert_allclose(got["ctrs"], ref.ctrs, rtol=0, atol=1e-13)
    # the eigen-free node path and the reference route (eigh on every tree node) agree on this live set
    os.environ["DH_REBUILD_FAST"] = "0"
    try:
        slow = ctx.rebuild(pts, multi=True)
    finally:
        del os.environ["DH_REBUILD_FAST"]
    assert slow["nells"] == m
    np.testing.assert_allclose(slow["covs"], got["covs"], rtol=1e-10, atol=0)
    np.testing.assert_allclose(slow["axes"], got["axes"], rtol=0, atol=1e-9 * np.abs(got["axes"]).max())


def test_rebuild_is_bit_reproducible_under_stress(ctx):
    """The parts of a k-means node exchange partial sums through agent-scope stores and a counter barrier.  Round 2
    found that __syncthreads() does not wait for those stores (s_waitcnt lgkmcnt(0); s_barrier), so a partner could
    read the previous iteration's partial: 3 of 2 700 eggbox rebuilds differed (different clusters, not rounding)
    once 128-point parts made multi-part nodes common.  The barrier now drains the stores explicitly; this test
    repeats batched rebuilds of captured eggbox / C2 live sets (permuted) and requires identical bits every time."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "livesets.npz"))
    fields = ("ctrs", "covs", "ams", "axes", "axlens", "logvol_ells")
    for name, reps in (("C3", 60), ("C2", 20)):
        rng = np.random.default_rng(0)
        sets = [g[f"{name}/{i}/live_u"][rng.permutation(len(g[f"{name}/{i}/live_u"]))] for i in range(3) for _ in range(6)]
        ref = ctx.rebuild_many(sets, multi=True)
        for _ in range(reps):
            got = ctx.rebuild_many(sets, multi=True)
            for x, y in zip(ref, got):
                assert x["nells"] == y["nells"]
                for k in fields:
                    np.testing.assert_array_equal(x[k], y[k])
