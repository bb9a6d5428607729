"""Wide-D path (wave-per-walker proposals, 1024-thread single-ellipsoid
rebuild, workgroup-per-point membership): BASELINE config C4 shapes (200-D,
N=4000, single/rslice) and dimensions without a register-resident
instantiation, against the oracle on the same seeds."""
import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B
from oracle import proposals_ref as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def test_eval_and_contains_200d(ctx):
    prob = inputs.problem("C4")
    rng = np.random.default_rng(2)
    u = rng.uniform(0.01, 0.99, size=(70, 200))
    v, logl = ctx.problem_eval(prob, u)
    np.testing.assert_allclose(v, prob.prior_transform_many(u), rtol=1e-11,
                               atol=1e-12)
    np.testing.assert_allclose(
        logl, prob.loglikelihood_many(prob.prior_transform_many(u)), rtol=1e-11)
    d, m = 200, 2
    ctrs = rng.uniform(0.4, 0.6, size=(m, d))
    a = rng.standard_normal((m, d, d)) * 0.05
    ams = np.einsum('mij,mkj->mik', a, a) + 3 * np.eye(d)
    x = ctrs[0] + 0.05 * rng.standard_normal((130, d))
    count, mask, quad = ctx.contains(x, ctrs, ams, want_mask=True,
                                     want_quad=True)
    want = np.array([B.multi_quadforms(p, ctrs, ams) for p in x])
    np.testing.assert_allclose(quad, want, rtol=1e-11)
    np.testing.assert_array_equal(count, (want < 1).sum(1))
    bits = np.unpackbits(mask.view(np.uint8).reshape(m, -1), axis=1,
                         bitorder="little")[:, :130].astype(bool)
    np.testing.assert_array_equal(bits, (want < 1).T)


@pytest.mark.parametrize("split", [True, False])
def test_single_rebuild_200d(ctx, split, monkeypatch):
    """Ellipsoid.update on the C4-shaped live set (4000 x 200): the multi-workgroup sequence
    (partial covariances, one-sided block Jacobi, partial Mahalanobis maxima) and the single-launch
    form with the two-sided solver (DH_WIDE_EIG=0; also the fallback when the eigensolver's
    workgroups cannot all be resident)."""
    if not split:
        monkeypatch.setenv("DH_WIDE_EIG", "0")
    pts = inputs.cloud("g200")
    got = ctx.rebuild(pts, multi=False)
    ref = B.bounding_ellipsoid(pts)
    np.testing.assert_allclose(got["ctrs"][0], ref.ctr, rtol=0, atol=1e-13)
    np.testing.assert_allclose(got["covs"][0], ref.cov, rtol=1e-9,
                               atol=1e-9 * np.abs(ref.cov).max())
    np.testing.assert_allclose(got["ams"][0], ref.am, rtol=0,
                               atol=1e-8 * np.abs(ref.am).max())
    np.testing.assert_allclose(got["logvol_ells"][0], ref.logvol, atol=1e-8)
    np.testing.assert_allclose(got["axlens"][0], np.sort(ref.axlens),
                               rtol=1e-9)
    ax = got["axes"][0]
    np.testing.assert_allclose(ax @ ax.T, ref.cov, rtol=0,
                               atol=1e-9 * np.abs(ref.cov).max())
    d = pts - got["ctrs"][0]
    q = np.einsum('ij,jk,ik->i', d, got["ams"][0], d)
    np.testing.assert_allclose(q.max(), 1 - 1e-3, rtol=1e-8)
    # MultiEllipsoid.update on the same cloud: 4000 points in 200-D never split (one Gaussian blob)
    multi = ctx.rebuild(pts, multi=True)
    assert multi["nells"] == 1
    np.testing.assert_allclose(multi["logvol_ells"][0], ref.logvol, rtol=1e-9)


@pytest.mark.parametrize("pname,d", [("C4", 200), ("N7", 7)])
def test_rslice_wide_vs_oracle(ctx, pname, d):
    from dynesty_amd import _lib, problems
    prob = inputs.problem("C4") if pname == "C4" else \
        problems.gauss_normal_prior(7, "N7")
    rng = np.random.default_rng(4)
    k = 6
    u0 = np.clip(0.5 + 0.08 * rng.standard_normal((k, d)), 0.02, 0.98)
    logl0 = prob.loglikelihood_many(prob.prior_transform_many(u0))
    loglstar = float(logl0.min() - 5.0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    axes = q * (0.08 * np.sqrt(d) * rng.uniform(0.8, 1.4, size=d))
    ent = [31, 32]
    st = ctx.seed_children(ent, 0, k)
    slices = 3
    out = ctx.slice_batch(prob, u0, axes, 0.8, loglstar, slices, st)
    kids = np.random.SeedSequence(ent).spawn(k)
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        ref = P.rslice(u0[i].copy(), loglstar, axes, 0.8, prob.prior_transform,
                       prob.loglikelihood, np.random.Generator(bg), slices)
        assert ref["ncalls"] == out["ncalls"][i]
        assert ref["n_expand"] == out["n_expand"][i]
        assert ref["n_contract"] == out["n_contract"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["logl"][i], ref["logl"], rtol=1e-10)
        np.testing.assert_array_equal(out["rng_out"][i],
                                      _lib.pcg_state_words(bg))


def test_rwalk_and_unitcube_wide_vs_oracle(ctx):
    from dynesty_amd import _lib, problems
    prob = problems.gauss_corr(40, 0.3, 5.0, "G40")  # precision-matrix path, D > 32
    d = 40
    rng = np.random.default_rng(6)
    k = 5
    u0 = np.clip(0.5 + 0.05 * rng.standard_normal((k, d)), 0.02, 0.98)
    loglstar = float(prob.loglikelihood_many(
        prob.prior_transform_many(u0)).min() - 8.0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    axes = q * (0.05 * np.sqrt(d))
    ent = [41]
    st = ctx.seed_children(ent, 0, k)
    out = ctx.rwalk_batch(prob, u0, axes, 0.6, loglstar, 30, st)
    kids = np.random.SeedSequence(ent).spawn(k)
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        ref = P.rwalk(u0[i].copy(), loglstar, axes, 0.6, prob.prior_transform,
                      prob.loglikelihood, np.random.Generator(bg), 30)
        assert ref["accept"] == out["accept"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(out["logl"][i], ref["logl"], rtol=1e-10)
        np.testing.assert_array_equal(out["rng_out"][i],
                                      _lib.pcg_state_words(bg))
    # unit cube in 40-D with a very low threshold
    st = ctx.seed_children([43], 0, 4)
    out = ctx.unif_batch(prob, -1e6, st)
    kids = np.random.SeedSequence([43]).spawn(4)
    for i in range(4):
        ref = P.unitcube(-1e6, prob.prior_transform, prob.loglikelihood,
                         np.random.Generator(np.random.PCG64(kids[i])), d)
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=0)
        assert out["ncalls"][i] == ref["ncalls"]


def test_rslice_logz_matches_the_reference_runs(ctx):
    """End-to-end statistical parity on the C4 family (D = 64: the wide kernels, bound='single',
    sample='rslice'): the reference's own logZ sits +0.37 above the analytic value at these
    settings (tests/golden/rslice_bias_ref.json, three seeds of the real dynesty) -- a property of
    the sampler settings.  The device runs must land on the REFERENCE's value: difference of the
    means within 3 standard errors."""
    import json
    import os
    from dynesty_amd import nested, problems
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rslice_bias_ref.json")))["runs"]
    prob = problems.gauss_normal_prior(64, "C4")
    ours = [nested.run_static(prob, nlive=500, bound='single', sample='rslice', queue_size=100,
                              rstate=np.random.default_rng(s), dlogz=0.01) for s in (11, 12, 13)]
    zr = np.array([r["logz"] for r in ref])
    zo = np.array([r.logz for r in ours])
    err = np.mean([r["logzerr"] for r in ref])
    se = err * np.sqrt(1.0 / len(zr) + 1.0 / len(zo))
    assert abs(zo.mean() - zr.mean()) < 3.0 * se, (zo, zr, se)
    assert abs(np.mean([r.logzerr for r in ours]) - err) < 0.02


@pytest.mark.parametrize("d,sizes", [(48, (2000, 2000)), (64, (2000, 1800, 2200)), (50, (900,)), (96, (2500, 2400))])
def test_multi_rebuild_wide_vs_oracle(ctx, d, sizes):
    """MultiEllipsoid.update above the register-resident limit (D > 44): host recursion over device
    node work (k-means in one workgroup, ellipsoids by the multi-workgroup rebuild).  Against the
    oracle's split tree: same number of ellipsoids, the same point partition, the same ellipsoids."""
    pts = inputs.blobs(d, sizes, 0.3, d)
    trace = []
    first = B.bounding_ellipsoid(pts)
    ells = B.split_tree(pts, first, trace=trace)
    got = ctx.rebuild(pts, multi=True, want_labels=True)
    assert got["nells"] == len(ells) == len(sizes)
    lab = got["labels"]
    # list order follows the sign of the major-axis eigenvector (LAPACK's is implementation-defined,
    # ours canonical): match the ellipsoids by centre, as tests/test_gpu_rebuild.py does
    octr = np.array([e.ctr for e in ells])
    order = [int(np.argmin(np.linalg.norm(octr - c, axis=1))) for c in got["ctrs"]]
    assert sorted(order) == list(range(len(ells)))
    for i, j in enumerate(order):
        e = ells[j]
        mine = pts[lab == i]
        np.testing.assert_allclose(mine.mean(axis=0), e.ctr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(got["ctrs"][i], e.ctr, rtol=0, atol=1e-12)
        np.testing.assert_allclose(got["logvol_ells"][i], e.logvol, rtol=0, atol=1e-8)
        np.testing.assert_allclose(got["covs"][i], e.cov, rtol=0, atol=1e-9 * np.abs(e.cov).max())
        np.testing.assert_allclose(got["ams"][i], e.am, rtol=0, atol=1e-8 * np.abs(e.am).max())
        ax = got["axes"][i]
        np.testing.assert_allclose(ax @ ax.T, e.cov, rtol=0, atol=1e-9 * np.abs(e.cov).max())
    # all points inside the union
    cnt, _, _ = ctx.contains(pts, got["ctrs"], got["ams"])
    assert np.all(cnt >= 1)


@pytest.mark.parametrize("case", ["single", "multi", "nc40"])
def test_unif_wide_vs_oracle(ctx, case):
    """UniformBoundSampler inside a (multi-)ellipsoid above the register-resident limit (D = 48):
    wave-per-walker kernel against the oracle on the same child streams -- accepted points, call
    counts and final generator words; the lock-step form (problem = -1) proposes the same points."""
    from oracle_backend import OracleBackend
    from dynesty_amd import problems
    d = 48
    prob = problems.gauss_normal_prior(d, "C4")
    rng = np.random.default_rng(3)
    nc = 40 if case == "nc40" else d
    p1 = 0.5 + 0.02 * rng.standard_normal((600, nc))
    p2 = 0.5 + 0.012 + 0.02 * rng.standard_normal((600, nc))
    e1, e2 = B.bounding_ellipsoid(p1), B.bounding_ellipsoid(p2)
    # threshold from points distributed like the proposals (the last d - nc coordinates are U(0, 1))
    _, ll = ctx.problem_eval(prob, np.hstack([p1, rng.random((600, d - nc))]))
    # (uniform draws in a 48-D ellipsoid sit near its surface: a threshold inside the cloud would
    # never be met -- take one below the cloud, 1/q and unitcheck still decide what is evaluated)
    loglstar = float(np.min(ll)) - 3.0
    if case == "multi":
        kw = dict(ctrs=np.array([e1.ctr, e2.ctr]), axes=np.array([e1.axes, e2.axes]),
                  ams=np.array([e1.am, e2.am]), logvol_ells=np.array([e1.logvol, e2.logvol]))
    else:
        kw = dict(ctrs=e1.ctr, axes=e1.axes, ncdim=nc)
    st = ctx.seed_children([9, 8, 7], 0, 24)
    got = ctx.unif_batch(prob, loglstar, st, max_tries=100000, **kw)
    ref = OracleBackend().unif_batch(prob, loglstar, st, **kw)
    np.testing.assert_array_equal(got["ncalls"], ref["ncalls"])
    np.testing.assert_array_equal(got["rng_out"], ref["rng_out"])
    np.testing.assert_allclose(got["u"], ref["u"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(got["logl"], ref["logl"], rtol=1e-10, atol=1e-9)
    assert np.all(got["logl"] > loglstar)
    if case == "multi":
        # lock-step form: the first proposal of every stream equals the oracle's first candidate
        pk = dict(ctrs=kw["ctrs"], axes=kw["axes"], ams=kw["ams"], logvol_ells=kw["logvol_ells"], ncdim=d)
        up, out = ctx.unif_propose(d, st, **pk)
        up2, out2 = OracleBackend().unif_propose(d, st, **pk)
        np.testing.assert_array_equal(out, out2)
        np.testing.assert_allclose(up, up2, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("which", ["rslice", "slice", "rwalk", "rwalk_nc"])
def test_wide_walkers_on_different_frames(ctx, which):
    """Walkers of one workgroup on DIFFERENT proposal frames (multi-ellipsoid bound: axes_idx) take
    the per-wavefront form of the frame product instead of the workgroup GEMM; principal-axes
    slice sampling and ncdim < ndim at wide D.  Against the oracle, walker by walker."""
    from oracle_backend import OracleBackend
    from dynesty_amd import problems
    d = 50
    prob = problems.gauss_normal_prior(d, "C4")
    rng = np.random.default_rng(10)
    k = 7
    u0 = np.clip(0.5 + 0.06 * rng.standard_normal((k, d)), 0.02, 0.98)
    loglstar = float(prob.loglikelihood_many(prob.prior_transform_many(u0)).min() - 6.0)
    nc = 44 if which == "rwalk_nc" else d
    frames = []
    for s in (0.05, 0.08):
        q, _ = np.linalg.qr(rng.standard_normal((nc, nc)))
        frames.append(q * (s * np.sqrt(nc) * rng.uniform(0.8, 1.3, size=nc)))
    frames = np.array(frames)
    idx = np.array([0, 1, 1, 0, 1, 0, 0], dtype=np.int32)
    st = ctx.seed_children([77, 1], 0, k)
    ob = OracleBackend()
    if which in ("rslice", "slice"):
        kw = dict(principal=(which == "slice"), axes_idx=idx)
        nsl = 2 if which == "rslice" else 1
        got = ctx.slice_batch(prob, u0, frames, 0.9, loglstar, nsl, st, **kw)
        ref = ob.slice_batch(prob, u0, frames, 0.9, loglstar, nsl, st, **kw)
        for key in ("ncalls", "n_expand", "n_contract"):
            np.testing.assert_array_equal(got[key], ref[key])
    else:
        kw = dict(axes_idx=idx, ncdim=nc)
        got = ctx.rwalk_batch(prob, u0, frames, 0.7, loglstar, 20, st, **kw)
        ref = ob.rwalk_batch(prob, u0, frames, 0.7, loglstar, 20, st, **kw)
        np.testing.assert_array_equal(got["accept"], ref["accept"])
    np.testing.assert_array_equal(got["rng_out"], ref["rng_out"])
    np.testing.assert_allclose(got["u"], ref["u"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(got["logl"], ref["logl"], rtol=1e-10)


@pytest.mark.parametrize("d,n,kind", [(45, 300, "gauss"), (77, 400, "gauss"), (128, 700, "corr"), (333, 1200, "gauss"),
                                      (512, 1600, "corr"), (60, 400, "tiny"), (64, 500, "flat"), (50, 40, "few")])
def test_single_rebuild_wide_dimensions(ctx, d, n, kind):
    """Ellipsoid.update across the wide range (block sizes / workgroup counts of the eigensolver
    change with D), strongly correlated clouds, a cloud of scale 1e-12 (the eigensolver works on
    the matrix scaled to max |a_ij| = 1), a rank-deficient cloud and one with fewer points than
    dimensions (both take improve_covar_mat's regularisation loop, i.e. the single-workgroup
    continuation with the two-sided solver)."""
    rng = np.random.default_rng(d + n)
    if kind == "gauss":
        pts = 0.5 + 0.03 * rng.standard_normal((n, d))
    elif kind == "corr":
        L = rng.standard_normal((d, d)) * 0.3 + np.eye(d)
        pts = 0.5 + 0.01 * rng.standard_normal((n, d)) @ L
    elif kind == "tiny":
        pts = 0.5 + 1e-12 * rng.standard_normal((n, d))
    elif kind == "flat":
        pts = 0.5 + 0.03 * rng.standard_normal((n, 20)) @ rng.standard_normal((20, d))
    else:
        pts = 0.5 + 0.03 * rng.standard_normal((n, d))
    got = ctx.rebuild(pts, multi=False)
    ref = B.bounding_ellipsoid(pts)
    loose = kind in ("flat", "few")
    np.testing.assert_allclose(got["ctrs"][0], ref.ctr, rtol=0, atol=1e-13)
    scale = np.abs(ref.cov).max()
    # 'tiny': deviations of 1e-12 around 0.5 carry four digits; the covariance is determined to ~1e-3
    ctol = 5e-3 if kind == "tiny" else (1e-5 if loose else 1e-9)
    np.testing.assert_allclose(got["covs"][0], ref.cov, rtol=0, atol=ctol * scale)
    np.testing.assert_allclose(got["logvol_ells"][0], ref.logvol, rtol=0,
                               atol=0.2 if kind == "tiny" else (1e-2 if loose else 1e-7))
    ax = got["axes"][0]
    np.testing.assert_allclose(ax @ ax.T, got["covs"][0], rtol=0, atol=(1e-6 if loose else 1e-10) * scale)
    if not loose:
        np.testing.assert_allclose(got["ams"][0] @ got["covs"][0], np.eye(d), rtol=0, atol=1e-7)
    dlt = pts - got["ctrs"][0]
    q = np.einsum('ij,jk,ik->i', dlt, got["ams"][0], dlt)
    assert q.max() <= 1.0


def _wide_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "wide.npz"))


@pytest.mark.parametrize("d,sizes", [(48, (2000, 2000)), (64, (2000, 1800, 2200))])
def test_multi_rebuild_wide_golden(ctx, d, sizes):
    """Device MultiEllipsoid.update above D = 44 against the REAL reference's ellipsoids
    (tests/golden/wide.npz), matched by centre."""
    g = _wide_golden()
    got = ctx.rebuild(inputs.blobs(d, sizes, 0.3, d), multi=True)
    assert got["nells"] == int(g[f"multi{d}/nells"])
    rc = g[f"multi{d}/ctrs"]
    for i in range(got["nells"]):
        j = int(np.argmin(np.linalg.norm(rc - got["ctrs"][i], axis=1)))
        np.testing.assert_allclose(got["ctrs"][i], rc[j], rtol=0, atol=1e-12)
        np.testing.assert_allclose(got["logvol_ells"][i], g[f"multi{d}/logvol_ells"][j], rtol=0, atol=1e-8)
        cov = g[f"multi{d}/covs"][j]
        np.testing.assert_allclose(got["covs"][i], cov, rtol=0, atol=1e-9 * np.abs(cov).max())


def test_single_rebuild_200d_golden(ctx):
    g = _wide_golden()
    got = ctx.rebuild(inputs.cloud("g200"), multi=False)
    np.testing.assert_allclose(got["ctrs"][0], g["single200/ctr"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(got["logvol_ells"][0], float(g["single200/logvol"]), rtol=0, atol=1e-7)
    np.testing.assert_allclose(np.sort(got["axlens"][0]), g["single200/axlens_sorted"], rtol=1e-9)
    rows = g["single200/cov_rows"]
    np.testing.assert_allclose(got["covs"][0][[0, 17]], rows, rtol=0, atol=1e-9 * np.abs(rows).max())
    np.testing.assert_allclose(np.trace(got["covs"][0]), float(g["single200/cov_trace"]), rtol=1e-10)


@pytest.mark.parametrize("tag,d", [("rslice64", 64), ("rwalk50", 50)])
def test_wide_walkers_golden(ctx, tag, d):
    """Wide walk kernels against the REAL reference's RSliceSampler.sample / RWalkSampler.sample on
    the same child streams: counters exact, coordinates to 1e-11 (frame product and norm are summed
    in a different order)."""
    g = _wide_golden()
    case = inputs.wide_walker_case(d, 6, d)
    st = ctx.seed_children([int(g[f"{tag}/seedbase"])], 0, 6)
    if tag.startswith("rslice"):
        out = ctx.slice_batch(case["problem"], case["u0"], case["axes"], 0.8, case["loglstar"], 3, st)
        np.testing.assert_array_equal(out["ncalls"], g[f"{tag}/ncalls"])
        np.testing.assert_array_equal(out["n_expand"], g[f"{tag}/ti_n_expand"])
        np.testing.assert_array_equal(out["n_contract"], g[f"{tag}/ti_n_contract"])
    else:
        out = ctx.rwalk_batch(case["problem"], case["u0"], case["axes"], 0.8, case["loglstar"], 30, st)
        np.testing.assert_array_equal(out["accept"], g[f"{tag}/ti_accept"])
    np.testing.assert_allclose(out["u"], g[f"{tag}/u"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(out["logl"], g[f"{tag}/logl"], rtol=1e-10)


def test_device_resident_loop_above_the_register_dimensions(ctx):
    """dh_ns_ensemble at D = 64 (wave-per-walker kernels with per-run thresholds / scales, masked multi-workgroup
    Ellipsoid.update): the same reference pin as the host-driven loop above (rslice_bias_ref.json: the real
    dynesty at nlive 500, bound='single', sample='rslice'), determinism, and a run's result independent of its
    shard mates (run i of an ensemble == the same run alone via first_run)."""
    import json
    import os
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rslice_bias_ref.json")))["runs"]
    prob = problems.gauss_normal_prior(64, "C4")
    kw = dict(bound='single', sample='rslice', dlogz=0.01, entropy=[5], max_iter=60000)
    r = ctx.ns_ensemble(prob, 6, 500, 100, **kw)
    assert (r["status"] == 0).all() and (r["nbound"] > 10).all()
    zr = np.array([x["logz"] for x in ref])
    err = np.mean([x["logzerr"] for x in ref])
    se = err * np.sqrt(1.0 / len(zr) + 1.0 / 6)
    assert abs(r["logz"].mean() - zr.mean()) < 3.0 * se, (r["logz"], zr, se)
    assert abs(r["logzerr"].mean() - err) < 0.02
    again = ctx.ns_ensemble(prob, 6, 500, 100, **kw)
    np.testing.assert_array_equal(again["logz"], r["logz"])
    alone = ctx.ns_ensemble(prob, 2, 500, 100, first_run=3, **kw)
    np.testing.assert_array_equal(alone["logz"], r["logz"][3:5])
    np.testing.assert_array_equal(alone["ncall"], r["ncall"][3:5])


@pytest.mark.parametrize("d,bound,sample", [(40, "multi", "rwalk"), (48, "single", "rwalk"), (7, "multi", "rslice"),
                                           (7, "single", "slice")])
def test_device_resident_loop_mixed_paths(ctx, d, bound, sample):
    """Dimensions where the loop mixes the paths: 33..44 = narrow (multi-ellipsoid) rebuild + wide walkers; above 44
    = wide rebuild + wide walkers; slice samplers at a dimension without a register-resident instantiation = narrow
    rebuild + wide walkers.  ln Z against the analytic value of the iid-Normal / Normal-prior family."""
    from dynesty_amd import backend, nested, problems
    prob = problems.gauss_normal_prior(d, "C4")
    r = ctx.ns_ensemble(prob, 8, 400, 64, bound=bound, sample=sample, dlogz=0.05, entropy=[d], max_iter=60000)
    assert (r["status"] == 0).all(), r["status"]
    lz = r["logz"]
    se = lz.std(ddof=1) / np.sqrt(8)
    # the quoted error against the scatter of the eight runs: (n - 1) s^2 / sigma^2 is chi-square with 7 degrees of
    # freedom, 0.1 % .. 99.9 % = 0.60 .. 24.3, i.e. sigma / s in 0.54 .. 3.4.  Upper side: that bound (round 6: the former
    # |ratio - 1| < 0.8 put a 5 % tail there and met it when the block eigensolver changed the last bits of a few
    # declined nodes).  Lower side: 0.2 as before -- rwalk at these dimensions under-mixes (below), so the runs scatter
    # by more than the quoted error (ratio 0.42 at d = 48).
    assert 0.2 < r["logzerr"].mean() / lz.std(ddof=1) < 3.4
    if sample == "rwalk":
        # rwalk with the default walks = 20 + d under-mixes at these dimensions (ln Z comes out +1.5 high at d = 48,
        # in the host-driven loop over the single-launch kernels exactly as here): hold the resident loop to that
        # loop -- same kernels, independent loop bookkeeping -- not to the analytic value
        backend.set_backend(ctx)
        try:
            host = np.array([nested.run_static(prob, nlive=400, bound=bound, sample=sample, queue_size=64,
                                               rstate=np.random.default_rng(100 + i), dlogz=0.05).logz for i in range(8)])
        finally:
            backend.set_backend(None)
        se2 = np.hypot(se, host.std(ddof=1) / np.sqrt(8))
        assert abs(lz.mean() - host.mean()) < 4 * se2, (lz.mean(), host.mean(), se2)
    else:
        assert abs(lz.mean() - prob.logz_truth) < 5 * se + 0.1, (lz.mean(), prob.logz_truth, se)


@pytest.mark.parametrize("bound,D", [("single", 36), ("multi", 34)])
def test_device_resident_loop_uniform_sampler_above_32(ctx, bound, D):
    """sample='unif' in the resident loop above the register-resident dimensions (round 5: the ensemble form of
    wide_unif_kernel -- per-run thresholds and bounds; round 4 refused it).  Event for event against the host mirror,
    which draws through the single-call entry point: the same deaths in the same slots, the same calls and bound
    updates; and a run does not depend on its shard mates.  (A wide Normal likelihood under a narrow uniform prior
    keeps the uniform sampler's efficiency bearable at this dimension.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from resident_mirror import mirror_run
    from dynesty_amd import problems
    prob = problems.gauss_iid(D, 1.2, f"unif{D}")
    nlive, K, dlogz, ent = 120, 8, 2.0, [D, 3]
    # (the bound phase from 300 calls on whatever the unit cube's efficiency)
    opt = dict(first_update=dict(min_ncall=300, min_eff=100.0))
    kw = dict(bound=bound, sample='unif', dlogz=dlogz, entropy=ent, rebuild_every=1, bootstrap=0, enlarge=1.25,
              want_samples=True, want_dead_logl=True, max_iter=4000, **opt)
    r = ctx.ns_ensemble(prob, 3, nlive, K, **kw)
    assert (r["status"] == 0).all()
    for run in (0, 2):
        m = mirror_run(ctx, prob, nlive, K, 1, bound, ent, run, dlogz, enlarge=1.25, sample="unif", bootstrap=0, **opt)
        n = int(r["niter"][run])
        assert m["done"] and m["niter"] == n, (m["niter"], n)
        np.testing.assert_array_equal(r["dead_id"][run, :n], np.array(m["dead_slot"]))
        np.testing.assert_allclose(r["dead_logl"][run, :n], np.array(m["dead_logl"]), rtol=1e-6, atol=0)
        assert int(r["ncall"][run]) == m["ncall"] and int(r["nbound"][run]) == m["nbound"]
        assert m["nbound"] >= 2
    alone = ctx.ns_ensemble(prob, 1, nlive, K, first_run=2, **kw)
    np.testing.assert_array_equal(alone["logz"], r["logz"][2:3])
    np.testing.assert_array_equal(alone["ncall"], r["ncall"][2:3])


def test_device_resident_loop_multi_bound_above_44(ctx):
    """bound='multi' above D = 44 in the resident loop: the masked wide MultiEllipsoid.update (host recursion over
    device node work, run mask read back on rebuild fills).  On a unimodal cloud the tree keeps its root, the same
    ellipsoid as bound='single' up to the eigensolver's start (warm there, cold per node here), so the two ensembles
    run the same streams and land within rounding-driven differences of each other; a run's result must not depend
    on its shard mates; ln Z agrees with the single-bound ensemble and the analytic value."""
    from dynesty_amd import problems
    prob = problems.gauss_normal_prior(48, "C4")
    kw = dict(sample='rslice', dlogz=0.05, entropy=[48], max_iter=60000)
    m = ctx.ns_ensemble(prob, 6, 400, 64, bound='multi', **kw)
    s1 = ctx.ns_ensemble(prob, 6, 400, 64, bound='single', **kw)
    assert (m["status"] == 0).all() and (m["nbound"] > 5).all(), (m["status"], m["nbound"])
    se = np.hypot(m["logz"].std(ddof=1), s1["logz"].std(ddof=1)) / np.sqrt(6)
    assert abs(m["logz"].mean() - s1["logz"].mean()) < 4 * se + 1e-12
    assert abs(m["logz"].mean() - prob.logz_truth) < 5 * m["logz"].std(ddof=1) / np.sqrt(6) + 0.1
    alone = ctx.ns_ensemble(prob, 2, 400, 64, bound='multi', first_run=2, **kw)
    np.testing.assert_array_equal(alone["logz"], m["logz"][2:4])
    np.testing.assert_array_equal(alone["ncall"], m["ncall"][2:4])


def test_device_resident_loop_bootstrap_above_44(ctx):
    """bootstrap > 0 above D = 44: the replicas of a rebuilding run go through the wide constructions one by one
    (rebuild_launch_full's ragged form above the narrow kernels).  The expansion only widens the bound: ln Z must agree
    with the plain run's ensemble and the analytic value, with more calls per iteration spent."""
    from dynesty_amd import problems
    prob = problems.gauss_normal_prior(48, "C4")
    kw = dict(bound='single', sample='rslice', dlogz=0.05, entropy=[7], max_iter=60000)
    b = ctx.ns_ensemble(prob, 4, 400, 64, bootstrap=3, **kw)
    p = ctx.ns_ensemble(prob, 4, 400, 64, **kw)
    assert (b["status"] == 0).all(), b["status"]
    se = np.hypot(b["logz"].std(ddof=1), p["logz"].std(ddof=1)) / 2.0
    assert abs(b["logz"].mean() - p["logz"].mean()) < 4 * se
    assert abs(b["logz"].mean() - prob.logz_truth) < 5 * b["logz"].std(ddof=1) / 2.0 + 0.1
    again = ctx.ns_ensemble(prob, 4, 400, 64, bootstrap=3, **kw)
    np.testing.assert_array_equal(again["logz"], b["logz"])
