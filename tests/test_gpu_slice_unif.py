"""K7 (rslice / slice incl. doubling), K8 (unif, unit cube) and the sequential
Bound.samples draw vs golden vectors from the real reference (same seeds).
Tolerances as in test_gpu_rwalk.py; every counter (ncalls, n_expand,
n_contract) must match exactly."""
import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu

ATOL_U = 1e-12
RTOL_L = 1e-11


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def states_for(ctx, g, tag):
    return ctx.seed_children([int(g[f"{tag}/seedbase"])], 0,
                             int(g[f"{tag}/nwalk"]))


def check(out, g, tag, counters=True):
    np.testing.assert_array_equal(out["ncalls"], g[f"{tag}/ncalls"])
    if counters:
        np.testing.assert_array_equal(out["n_expand"], g[f"{tag}/ti_n_expand"])
        np.testing.assert_array_equal(out["n_contract"],
                                      g[f"{tag}/ti_n_contract"])
    np.testing.assert_allclose(out["u"], g[f"{tag}/u"], rtol=0, atol=ATOL_U)
    scale_v = max(1.0, np.abs(g[f"{tag}/v"]).max())
    np.testing.assert_allclose(out["v"], g[f"{tag}/v"], rtol=0,
                               atol=ATOL_U * 20 * scale_v)
    np.testing.assert_allclose(out["logl"], g[f"{tag}/logl"], rtol=RTOL_L,
                               atol=1e-11)


@pytest.mark.parametrize("pname,slices,cseed", [("C3", 5, 955), ("G5", 4, 954),
                                                ("N6", 3, 953), ("C2", 3, 953)])
def test_rslice_golden(ctx, pname, slices, cseed, golden_proposals):
    g = golden_proposals
    tag = f"rslice/{pname}"
    case = inputs.walker_case(pname, 64, cseed)
    nw = int(g[f"{tag}/nwalk"])
    out = ctx.slice_batch(case["problem"], case["u0"][:nw], case["axes"],
                          case["scale"], case["loglstar"], slices,
                          states_for(ctx, g, tag))
    check(out, g, tag)
    assert not out["expansion_warning_set"].any()


@pytest.mark.parametrize("tag,slices,scale,dbl", [
    ("rslice/G5_dbl", 4, None, True), ("rslice/G5_tiny", 2, 0.02, False)])
def test_rslice_variants_golden(ctx, tag, slices, scale, dbl, golden_proposals):
    g = golden_proposals
    case = inputs.walker_case("G5", 64, 961)
    nw = int(g[f"{tag}/nwalk"])
    out = ctx.slice_batch(case["problem"], case["u0"][:nw], case["axes"],
                          scale or case["scale"], case["loglstar"], slices,
                          states_for(ctx, g, tag), doubling=dbl)
    check(out, g, tag)


@pytest.mark.parametrize("tag,pname,cseed,dbl", [
    ("slice/G5", "G5", 972, False), ("slice/E3", "E3", 972, False),
    ("slice/G5_dbl", "G5", 981, True)])
def test_pslice_golden(ctx, tag, pname, cseed, dbl, golden_proposals):
    g = golden_proposals
    case = inputs.walker_case(pname, 64, cseed)
    nw = int(g[f"{tag}/nwalk"])
    out = ctx.slice_batch(case["problem"], case["u0"][:nw], case["axes"],
                          case["scale"], case["loglstar"], 2,
                          states_for(ctx, g, tag), principal=True,
                          doubling=dbl)
    check(out, g, tag)


def test_rslice_many_vs_oracle(ctx):
    """Ragged batch, two frames, advanced generator state returned."""
    from dynesty_amd import _lib
    from oracle import proposals_ref as P
    case = inputs.walker_case("G5", 300, 78)
    prob = case["problem"]
    u0 = case["u0"][:150]
    k = u0.shape[0]
    axes2 = np.stack([case["axes"], 0.7 * case["axes"][::-1, ::-1].copy()])
    idx = (np.arange(k) % 2).astype(np.int32)
    ent = [9, 9, 9]
    st = ctx.seed_children(ent, 0, k)
    out = ctx.slice_batch(prob, u0, axes2, case["scale"], case["loglstar"], 3,
                          st, axes_idx=idx)
    kids = np.random.SeedSequence(ent).spawn(k)
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        ref = P.rslice(u0[i].copy(), case["loglstar"], axes2[idx[i]],
                       case["scale"], prob.prior_transform, prob.loglikelihood,
                       np.random.Generator(bg), 3)
        assert ref["ncalls"] == out["ncalls"][i]
        assert ref["n_expand"] == out["n_expand"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=ATOL_U)
        np.testing.assert_array_equal(out["rng_out"][i],
                                      _lib.pcg_state_words(bg))


def test_unif_golden(ctx, golden_proposals):
    g = golden_proposals
    # single ellipsoid, C1
    prob = inputs.problem("C1")
    e = B.bounding_ellipsoid(inputs.cloud("g3"))
    tag = "unif/C1_single"
    st = ctx.seed_children([7000], 0, 16)
    out = ctx.unif_batch(prob, float(g[f"{tag}/loglstar"]), st, ctrs=e.ctr,
                         axes=e.axes)
    check(out, g, tag, counters=False)
    # overlapping union of ellipsoids, 1/q rejection
    prob = inputs.problem("G5")
    pts = inputs.cloud("two5")
    m = B.multi_update(pts)
    m = B.scale_multi_to_logvol(m, m.logvol + 5 * np.log(3.0))
    tag = "unif/G5_multi"
    st = ctx.seed_children([7100], 0, 16)
    out = ctx.unif_batch(prob, float(g[f"{tag}/loglstar"]), st, ctrs=m.ctrs,
                         axes=np.array([el.axes for el in m.ells]), ams=m.ams,
                         logvol_ells=m.logvol_ells)
    check(out, g, tag, counters=False)
    # 3-D bound on a 5-D problem: the last two dims are U(0,1)
    b3 = B.bounding_ellipsoid(pts[:, :3].copy())
    tag = "unif/G5_nc3"
    st = ctx.seed_children([7200], 0, 8)
    out = ctx.unif_batch(prob, float(g[f"{tag}/loglstar"]), st, ctrs=b3.ctr,
                         axes=b3.axes, ncdim=3)
    check(out, g, tag, counters=False)
    # unit cube
    prob = inputs.problem("C1")
    st = ctx.seed_children([7300], 0, 8)
    out = ctx.unif_batch(prob, -60.0, st)
    np.testing.assert_array_equal(out["ncalls"], g["unitcube/C1/ncalls"])
    np.testing.assert_allclose(out["u"], g["unitcube/C1/u"], rtol=0, atol=0)
    np.testing.assert_allclose(out["logl"], g["unitcube/C1/logl"], rtol=RTOL_L)


@pytest.mark.parametrize("name", ["c2", "c3", "two5", "g3"])
def test_bound_draw_golden(ctx, name, golden_bounding):
    """MultiEllipsoid.samples / Ellipsoid.samples from one generator; uses the
    reference's own ellipsoids (axes included) so coordinates are comparable."""
    from dynesty_amd import _lib
    g = golden_bounding
    ctrs, ams = g[f"{name}/mu/ctrs"], g[f"{name}/mu/ams"]
    axes, lvs = g[f"{name}/mu/axes"], g[f"{name}/mu/logvol_ells"]
    bg = np.random.PCG64(7)
    st = _lib.pcg_state_words(bg)
    xs, idxs, qs, out = ctx.bound_draw(st, 40, ctrs, axes, ams, lvs)
    np.testing.assert_allclose(xs, g[f"{name}/mu/samples"], rtol=0, atol=1e-13)
    # the generator must have advanced exactly as numpy's did
    m = B.stack_ells([B.Ell(ctrs[i], g[f"{name}/mu/covs"][i], ams[i], axes[i],
                            g[f"{name}/mu/axlens"][i], float(lvs[i]))
                      for i in range(len(lvs))])
    rng = np.random.Generator(bg)
    for _ in range(40):
        B.multi_sample(m, rng)
    np.testing.assert_array_equal(out, _lib.pcg_state_words(bg))
    # single-ellipsoid bound
    e = B.bounding_ellipsoid(inputs.cloud(name))
    bg = np.random.PCG64(8)
    xs, _, _, _ = ctx.bound_draw(_lib.pcg_state_words(bg), 40, e.ctr, e.axes)
    np.testing.assert_allclose(xs, g[f"{name}/single/samples"], rtol=0,
                               atol=1e-13)


@pytest.mark.parametrize("kind", ["multi", "single", "cube", "balls"])
def test_unif_lockstep_equals_fused(ctx, kind):
    """The lock-step path for arbitrary Python likelihoods (candidates from
    dh_unif_batch with problem = -1, callbacks on the host) reproduces the fused
    kernel: same accepted points, same call counts, same final streams."""
    from dynesty_amd import samplers, backend, bounding
    from dynesty_amd.samplers import SamplerReturn  # noqa: F401
    import types
    prob = inputs.problem("G5")
    rng = np.random.default_rng(4)
    live = 0.5 + 0.05 * rng.standard_normal((400, 5))
    _, ll = ctx.problem_eval(prob, live)
    loglstar = float(np.sort(ll)[80])
    backend.set_backend(ctx)
    try:
        if kind == "multi":
            b = bounding.HipMultiEllipsoid(5)
            b.update(np.vstack([live[:200], live[200:] + 0.2]))
        elif kind == "single":
            b = bounding.HipEllipsoid(5)
            b.update(live)
        elif kind == "balls":
            b = bounding.HipRadFriends(5)
            b.update(live)
            b.ctrs = live
        else:
            b = None
        kids = np.random.SeedSequence(77).spawn(40)

        def mk(problem):
            kw = dict(bound=b, ndim=5, n_cluster=5, nonbounded=None, problem=problem)
            return [types.SimpleNamespace(u=None, loglstar=loglstar, axes=None, scale=1.0,
                                          prior_transform=prob.prior_transform,
                                          loglikelihood=prob.loglikelihood,
                                          rseed=np.random.Generator(np.random.PCG64(s)), kwargs=kw)
                    for s in kids]
        if kind == "cube":
            st = ctx.seed_children([1, 2, 3, 4], 0, 40)
            fused = ctx.unif_batch(prob, loglstar - 50.0, st)
            u, out = st.copy(), None
            todo = np.arange(40); states = np.array(st).reshape(40, 4); nc = np.zeros(40, int); got = np.zeros((40, 5))
            while len(todo):
                up, so = ctx.unif_propose(5, states[todo])
                states[todo] = so
                keep = []
                for j, i in enumerate(todo):
                    nc[i] += 1
                    if prob.loglikelihood(prob.prior_transform(up[j])) > loglstar - 50.0:
                        got[i] = up[j]
                    else:
                        keep.append(i)
                todo = np.array(keep, dtype=int)
            np.testing.assert_array_equal(got, fused["u"])
            np.testing.assert_array_equal(nc, fused["ncalls"])
            np.testing.assert_array_equal(states, fused["rng_out"])
            return
        a_f, a_l = mk(prob), mk(None)
        fused = samplers.run_unif(a_f)
        lock = samplers.run_unif(a_l)
        for f, l, af, al in zip(fused, lock, a_f, a_l):
            np.testing.assert_array_equal(f.u, l.u)
            assert f.ncalls == l.ncalls
            assert abs(f.logl - l.logl) < 1e-10
            assert af.rseed.random() == al.rseed.random()
    finally:
        backend.set_backend(None)


def test_slice_feed_equals_numpy_streams(ctx):
    """dh_slice_feed (directions / axis orders / uncommitted uniform lookahead for the lock-step
    slice path) against NumPy Generators on the same 6-word states: streams, permutations and
    uniforms bit-exact, directions to rounding (norm and dot-product order differ)."""
    from oracle_backend import OracleBackend
    from dynesty_amd import _lib
    ob = OracleBackend()
    for nd in (3, 25, 70):
        k = 9
        gens = [np.random.Generator(np.random.PCG64(500 + i)) for i in range(k)]
        for i, g in enumerate(gens):
            if i % 2:
                g.integers(7)  # buffered 32-bit half
        st = np.array([_lib.pcg_state6(g.bit_generator) for g in gens], dtype=np.uint64)
        rng = np.random.default_rng(nd)
        axes = rng.standard_normal((2, nd, nd))
        idx = (np.arange(k) % 2).astype(np.int32)
        cons = (np.arange(k) * 3).astype(np.int32)
        for kind, kw in (("direction", dict(axes=axes, axes_idx=idx, scale=0.7)), ("shuffle", {}),
                         ("advance", {})):
            s1, o1, l1 = ctx.slice_feed(kind, nd, st, consumed=cons, nlook=70, **kw)
            s2, o2, l2 = ob.slice_feed(kind, nd, st, consumed=cons, nlook=70, **kw)
            np.testing.assert_array_equal(s1, s2)
            np.testing.assert_array_equal(l1, l2)
            if kind == "direction":
                np.testing.assert_allclose(o1, o2, rtol=1e-12, atol=1e-14)
            elif kind == "shuffle":
                np.testing.assert_array_equal(o1, o2)
            st = s1


@pytest.mark.parametrize("which", ["rslice", "slice"])
def test_slice_lockstep_on_device_feed(ctx, which):
    """run_rslice / run_slice without a device problem (arbitrary Python likelihood) on the HIP
    backend against the same host state machine fed by NumPy streams."""
    import types
    from oracle_backend import OracleBackend
    from dynesty_amd import samplers, backend
    prob = inputs.problem("G5")
    rng = np.random.default_rng(8)
    us = 0.5 + 0.04 * rng.standard_normal((12, 5))
    _, ll = ctx.problem_eval(prob, us)
    loglstar = float(np.min(ll)) - 1.0
    axes = 0.1 * (np.eye(5) + 0.2 * rng.standard_normal((5, 5)))

    def mk():
        kids = np.random.SeedSequence(31).spawn(12)
        return [types.SimpleNamespace(u=us[i].copy(), loglstar=loglstar, axes=axes, scale=1.1,
                                      prior_transform=prob.prior_transform,
                                      loglikelihood=prob.loglikelihood, rseed=kids[i],
                                      kwargs=dict(slices=4, slice_doubling=False, nonperiodic=None))
                for i in range(12)]
    run = samplers.run_rslice if which == "rslice" else samplers.run_slice
    backend.set_backend(ctx)
    try:
        got = run(mk())
        backend.set_backend(OracleBackend())
        ref = run(mk())
    finally:
        backend.set_backend(None)
    for g, r in zip(got, ref):
        assert g.ncalls == r.ncalls and g.tuning_info == r.tuning_info
        np.testing.assert_allclose(g.u, r.u, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g.logl, r.logl, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("pname", ["C1", "C3", "G5", "C2", "C4nd13"])
def test_unit_cube_four_lanes_per_walker_equals_one_lane(pname, monkeypatch):
    """UnitCubeSampler.sample with four lanes per walker (tries 4 r + t of round r from generators jumped t n draws
    ahead; DH_CUBE_FORM=2) against one walker per lane (=1): the same points, log-likelihoods, call counts and
    generator end states, bit for bit -- also through whole resident runs, whose unit-cube phase takes it."""
    from dynesty_amd import _lib, problems
    prob = problems.gauss_iid(13, 4.0, "g13") if pname == "C4nd13" else inputs.problem(pname)
    ctxs = []
    for form in ("1", "2"):
        monkeypatch.setenv("DH_CUBE_FORM", form)
        ctxs.append(_lib.Context(0))
    k = 777
    states = ctxs[0].seed_children(np.array([5, 6, 7, 8]), 0, k)
    # a threshold a few per cent of the cube beats: tens of tries per walker
    u = np.random.default_rng(1).random((4000, prob.ndim))
    ll = prob.loglikelihood_many(prob.prior_transform_many(u))
    loglstar = float(np.quantile(ll, 0.97))
    a = ctxs[0].unif_batch(prob, loglstar, states)
    b = ctxs[1].unif_batch(prob, loglstar, states)
    for key in ("u", "v", "logl", "ncalls", "rng_out"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert a["ncalls"].max() > 40 and np.all(a["logl"] > loglstar)
    kw = dict(nlive=200, queue_size=48, walks=15, bound="single", entropy=[4, 4], dlogz=0.5)
    ra = ctxs[0].ns_ensemble(prob, 5, **kw)
    rb = ctxs[1].ns_ensemble(prob, 5, **kw)
    np.testing.assert_array_equal(ra["logz"], rb["logz"])
    np.testing.assert_array_equal(ra["ncall"], rb["ncall"])
