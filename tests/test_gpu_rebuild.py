"""K1-K4 rebuild kernel (dh_rebuild) vs golden vectors of the real reference
(MultiEllipsoid.update / bounding_ellipsoid on the seeded clouds) and vs the
oracle's split trace.

What is compared and how:
  * nells: exact.
  * the SET of ellipsoids: the k-means seeds are ctr -/+ major axis, and the
    sign of a LAPACK eigenvector is arbitrary, so the reference's child order
    (hence list order) is not defined by the algorithm; ellipsoids are matched
    by centre before comparing.
  * ctr / cov / logvol / sorted axlens: tight fp64 tolerances (below).
  * am: relative to its largest entry (it is the inverse of cov).
  * axes: through the invariants axes @ axes.T == cov and, column by column
    against the reference after fixing both signs, where the eigenvalue gap
    allows it.
  * cluster membership of every point: exact (as a partition).
"""
import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def match_by_centre(ctrs_a, ctrs_b):
    """Permutation p with ctrs_a[i] ~ ctrs_b[p[i]]."""
    m = len(ctrs_a)
    dist = np.linalg.norm(ctrs_a[:, None, :] - ctrs_b[None, :, :], axis=2)
    p = dist.argmin(axis=1)
    assert len(set(p.tolist())) == m, "ellipsoid centres do not match one-to-one"
    return p


def canon_sign(axes):
    out = axes.copy()
    for k in range(axes.shape[1]):
        i = np.argmax(np.abs(out[:, k]))
        if out[i, k] < 0:
            out[:, k] = -out[:, k]
    return out


def check_ell(got, i, ctr, cov, am, axes, axlens, logvol, loose=False):
    rt = 1e-4 if loose else RTOL
    np.testing.assert_allclose(got["ctrs"][i], ctr, rtol=0, atol=1e-13)
    np.testing.assert_allclose(got["covs"][i], cov, rtol=rt,
                               atol=rt * np.abs(cov).max())
    np.testing.assert_allclose(got["ams"][i], am, rtol=0,
                               atol=(1e-3 if loose else 1e-8) * np.abs(am).max())
    np.testing.assert_allclose(got["logvol_ells"][i], logvol, rtol=0,
                               atol=1e-3 if loose else 1e-9)
    np.testing.assert_allclose(np.sort(got["axlens"][i]), np.sort(axlens),
                               rtol=1e-4 if loose else RTOL)
    ax = got["axes"][i]
    np.testing.assert_allclose(ax @ ax.T, cov, rtol=0,
                               atol=(1e-4 if loose else 1e-10) * np.abs(cov).max())
    # our own conventions: ascending axis lengths, canonical signs
    assert np.all(np.diff(got["axlens"][i]) >= 0)
    np.testing.assert_array_equal(ax, canon_sign(ax))
    np.testing.assert_allclose(np.linalg.norm(ax, axis=0), got["axlens"][i],
                               rtol=1e-12)


@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_multi_update_golden(ctx, name, golden_bounding):
    g = golden_bounding
    pts = inputs.cloud(name)
    got = ctx.rebuild(pts, multi=True, want_labels=True)
    assert got["nells"] == int(g[f"{name}/mu/nells"])
    p = match_by_centre(got["ctrs"], g[f"{name}/mu/ctrs"])
    # flat10 is rank 3 in 10-D: improve_covar_mat clips the noise eigenvalues
    # to 1e-11 * max, so cov/am carry the 1e11 condition number times rounding
    loose = name == "flat10"
    for i in range(got["nells"]):
        j = p[i]
        check_ell(got, i, g[f"{name}/mu/ctrs"][j], g[f"{name}/mu/covs"][j],
                  g[f"{name}/mu/ams"][j], g[f"{name}/mu/axes"][j],
                  g[f"{name}/mu/axlens"][j], g[f"{name}/mu/logvol_ells"][j],
                  loose=loose)
    # cluster membership: the oracle's leaves, as a partition of the points.  The oracle's split
    # tree (same k-means, same accept tests as the reference) yields the point set of every leaf;
    # each device cluster must be exactly one of them.
    ref_leaves = oracle_leaf_partition(pts)
    assert len(ref_leaves) == got["nells"]
    lab = got["labels"]
    assert lab.min() >= 0 and lab.max() == got["nells"] - 1
    ref_sets = {frozenset(ix.tolist()) for ix in ref_leaves}
    assert len(ref_sets) == len(ref_leaves)
    for i in range(got["nells"]):
        mine = frozenset(np.flatnonzero(lab == i).tolist())
        assert mine in ref_sets, f"device cluster {i} is not a leaf of the oracle's tree"
        ref_sets.discard(mine)
    assert not ref_sets


def oracle_leaf_partition(pts):
    """Index sets of the leaves MultiEllipsoid.update keeps, from the oracle's recursion
    (oracle.bounding_ref.split_tree restated over index arrays: bounding.py:1464-1563)."""
    from scipy.cluster.vq import kmeans2
    from scipy.special import logsumexp
    import warnings

    def rec(idx, ell, scale):
        sub = pts[idx]
        n, d = sub.shape
        if n < 4 * d:
            return [(idx, ell)]
        p1, p2 = B.major_axis_endpoints(ell)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, labels = kmeans2(sub / scale, k=np.vstack((p1, p2)) / scale, iter=10, minit='matrix',
                                check_finite=False)
        parts = [idx[labels == k] for k in (0, 1)]
        if min(len(parts[0]), len(parts[1])) < 2 * d:
            return [(idx, ell)]
        kids = [B.bounding_ellipsoid(pts[p]) for p in parts]
        dec = (d * (d + 3)) // 2 * np.log(n) / n
        out = rec(parts[0], kids[0], scale) + rec(parts[1], kids[1], scale)
        if np.logaddexp(kids[0].logvol, kids[1].logvol) - ell.logvol < -dec:
            return out
        if logsumexp([e.logvol for _, e in out]) - ell.logvol < -dec * (len(out) - 1):
            return out
        return [(idx, ell)]
    leaves = rec(np.arange(len(pts)), B.bounding_ellipsoid(pts), pts.std(axis=0)[None, :])
    # the restatement must agree with the oracle proper
    ells = B.split_tree(pts, B.bounding_ellipsoid(pts))
    assert len(ells) == len(leaves)
    for (ix, e), e2 in zip(leaves, ells):
        np.testing.assert_array_equal(e.ctr, e2.ctr)
    return [ix for ix, _ in leaves]


@pytest.mark.parametrize("name", ["c2", "c3", "two5", "ring2"])
def test_axes_columns_vs_reference(ctx, name, golden_bounding):
    """Where eigenvalues are separated the axes agree column by column with the
    reference's (LAPACK) axes once both are sign-normalised."""
    g = golden_bounding
    got = ctx.rebuild(inputs.cloud(name), multi=True)
    p = match_by_centre(got["ctrs"], g[f"{name}/mu/ctrs"])
    nchecked = 0
    for i in range(got["nells"]):
        ref_ax = canon_sign(g[f"{name}/mu/axes"][p[i]])
        ref_len = g[f"{name}/mu/axlens"][p[i]]
        gaps = np.abs(np.subtract.outer(ref_len, ref_len))
        np.fill_diagonal(gaps, np.inf)
        for k in range(len(ref_len)):
            if gaps[k].min() > 1e-3 * ref_len[k]:
                np.testing.assert_allclose(got["axes"][i][:, k], ref_ax[:, k],
                                           rtol=0, atol=1e-9 * ref_len[k] /
                                           (gaps[k].min() / ref_len[k]))
                nchecked += 1
    assert nchecked > 0


@pytest.mark.parametrize("name", inputs.CLOUDS_SMALL)
def test_single_golden(ctx, name, golden_bounding):
    g = golden_bounding
    got = ctx.rebuild(inputs.cloud(name), multi=False)
    assert got["nells"] == 1
    check_ell(got, 0, g[f"{name}/be/ctr"], g[f"{name}/be/cov"],
              g[f"{name}/be/am"], g[f"{name}/be/axes"], g[f"{name}/be/axlens"],
              float(g[f"{name}/be/logvol"]), loose=(name == "flat10"))


def test_all_points_covered_and_scaled(ctx):
    """fmax scaling: every point strictly inside, the outermost one at 1-1e-3."""
    for name in ("c2", "g3"):
        pts = inputs.cloud(name)
        got = ctx.rebuild(pts, multi=False)
        d = pts - got["ctrs"][0]
        q = np.einsum('ij,jk,ik->i', d, got["ams"][0], d)
        assert q.max() < 1.0
        np.testing.assert_allclose(q.max(), 1 - 1e-3, rtol=1e-9)


def test_errors(ctx):
    with pytest.raises(ValueError):
        ctx.rebuild(np.full((1, 3), 0.5), multi=False)
    with pytest.raises(RuntimeError):
        ctx.rebuild(np.full((1, 3), 0.5), multi=True)
    # identical points: covariance is exactly zero -> improve_covar_mat falls
    # back towards the identity, as the reference does (no exception)
    got = ctx.rebuild(np.full((40, 3), 0.5), multi=False)
    ref = B.bounding_ellipsoid(np.full((40, 3), 0.5))
    np.testing.assert_allclose(got["logvol_ells"][0], ref.logvol, atol=1e-9)


def test_number_of_clusters(ctx):
    """Reference tests/test_ellipsoid.py:267-286, scaled down: a 4^3 grid of
    tight Gaussian blobs must be resolved into ~one ellipsoid per blob."""
    rng = np.random.default_rng(1)
    nd, side, per = 3, 4, 40
    centres = np.stack(np.meshgrid(*[np.arange(side)] * nd), -1).reshape(-1, nd)
    pts = (centres[:, None, :] + 0.02 * rng.standard_normal(
        (len(centres), per, nd))).reshape(-1, nd)
    pts = (pts + 0.5) / side
    pts = pts[rng.permutation(len(pts))]
    got = ctx.rebuild(pts, multi=True, max_ells=256)
    want = len(B.multi_update(pts).ells)
    assert got["nells"] == want
    assert abs(got["nells"] - len(centres)) <= 0.1 * len(centres)


def test_ragged_batch_bootstrap(ctx, golden_bounding):
    """All bootstrap replicas in one ragged launch (dh_rebuild_ragged_dev) and
    the resulting expansion factors vs the reference's golden values
    (_ellipsoid_bootstrap_expand, bounding.py:1619-1648)."""
    from dynesty_amd import backend, bootstrap
    g = golden_bounding
    backend.set_backend(ctx)
    try:
        for name in ("g3", "two5", "c3"):
            pts = inputs.cloud(name)
            for multi, key in ((False, "single"), (True, "multi")):
                seeds = np.random.SeedSequence(77).spawn(3)
                splits = [bootstrap._split(pts, s) for s in seeds]
                res = ctx.rebuild_many([p for p, _ in splits], multi=multi)
                got = []
                for (pin, pout), r in zip(splits, res):
                    _, _, quad = ctx.contains(pout, r["ctrs"], r["ams"],
                                              want_quad=True)
                    got.append(max(1., float(np.sqrt(quad.min(axis=1)).max())))
                np.testing.assert_allclose(got, g[f"{name}/boot/{key}"],
                                           rtol=1e-8)
    finally:
        backend.set_backend(None)


def test_cooperative_launch_of_the_whole_grid_barrier_kernels(monkeypatch):
    """DH_COOP_LAUNCH=1: the kernels whose whole grid meets at spin barriers -- the root's parts (k_root_parts) and
    the multi-workgroup eigensolver of the wide bound (wide_eig2_kernel) -- go through hipLaunchCooperativeKernel,
    which has the runtime vouch for the co-residency the library otherwise derives from the occupancy query.  Same
    kernels, same results bit for bit (a 2000-point MultiEllipsoid.update, whose root has eight parts, and a 200-D
    Ellipsoid.update)."""
    from dynesty_amd import _lib
    rng = np.random.default_rng(5)
    pts = np.clip(0.5 + 0.1 * rng.standard_normal((2000, 25)), 0.001, 0.999)
    wide = np.clip(0.5 + 0.05 * rng.standard_normal((1200, 200)), 0.001, 0.999)
    outs = []
    for coop in ("0", "1"):
        monkeypatch.setenv("DH_COOP_LAUNCH", coop)
        c = _lib.Context(0)
        outs.append((c.rebuild(pts, multi=True), c.rebuild(wide, multi=False)))
        del c
    for a, b in zip(outs[0], outs[1]):
        assert a["nells"] == b["nells"]
        for key in ("ctrs", "covs", "ams", "axes", "logvol_ells"):
            np.testing.assert_array_equal(a[key], b[key])
