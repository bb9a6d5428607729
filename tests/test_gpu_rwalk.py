"""K6+K9 rwalk batch kernel vs golden vectors of the real reference and vs the
oracle on the same seeds.  Tolerances: u/v within 1e-12 absolute (unit-cube
coordinates / O(1) parameters), logl within 1e-11 relative, counts exact."""
import numpy as np
import pytest

import inputs
from oracle import proposals_ref as P

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


def check(out, g, tag):
    np.testing.assert_array_equal(out["accept"], g[f"{tag}/ti_accept"])
    np.testing.assert_array_equal(out["reject"], g[f"{tag}/ti_reject"])
    np.testing.assert_allclose(out["u"], g[f"{tag}/u"], rtol=0, atol=ATOL_U)
    scale_v = max(1.0, np.abs(g[f"{tag}/v"]).max())
    np.testing.assert_allclose(out["v"], g[f"{tag}/v"], rtol=0,
                               atol=ATOL_U * 20 * scale_v)
    np.testing.assert_allclose(out["logl"], g[f"{tag}/logl"], rtol=RTOL_L,
                               atol=1e-11)


RW = [("C2", 45, 945), ("G5", 25, 925), ("C1", 23, 923), ("C3", 22, 922),
      ("N6", 26, 926)]


@pytest.mark.parametrize("pname,walks,cseed", RW)
def test_rwalk_golden(ctx, pname, walks, cseed, golden_proposals):
    g = golden_proposals
    tag = f"rwalk/{pname}"
    case = inputs.walker_case(pname, 64, cseed)
    nw = int(g[f"{tag}/nwalk"])
    out = ctx.rwalk_batch(case["problem"], case["u0"][:nw], case["axes"],
                          case["scale"], case["loglstar"], walks,
                          states_for(ctx, g, tag))
    check(out, g, tag)


def test_rwalk_periodic_reflective_golden(ctx, golden_proposals):
    from dynesty_amd import _lib
    g = golden_proposals
    tag = "rwalk/G5_pr"
    case = inputs.walker_case("G5", 64, 931, shrink=3.0)
    bc = np.zeros(5, dtype=np.int8)
    bc[[0, 3]] = _lib.BC_PERIODIC
    bc[[1]] = _lib.BC_REFLECT
    out = ctx.rwalk_batch(case["problem"], case["u0"][:16], case["axes"], 2.5,
                          case["loglstar"], 30, states_for(ctx, g, tag), bc=bc)
    check(out, g, tag)


def test_rwalk_ncdim_golden(ctx, golden_proposals):
    g = golden_proposals
    tag = "rwalk/G5_nc3"
    case = inputs.walker_case("G5", 64, 932, shrink=2.0)
    out = ctx.rwalk_batch(case["problem"], case["u0"][:16],
                          case["axes"][:3, :3].copy(), case["scale"],
                          case["loglstar"], 20, states_for(ctx, g, tag),
                          ncdim=3)
    check(out, g, tag)


@pytest.mark.parametrize("pname,walks", [("C2", 45), ("G5", 25), ("C3", 30)])
def test_rwalk_vs_oracle_many(ctx, pname, walks):
    """A few hundred walkers incl. a ragged tail (k not a multiple of 64), two
    proposal frames selected per walker, rng_out = advanced numpy state."""
    from dynesty_amd import _lib
    case = inputs.walker_case(pname, 400, 77)
    prob = case["problem"]
    u0 = case["u0"][:333]
    k = u0.shape[0]
    axes2 = np.stack([case["axes"], 0.5 * case["axes"][::-1, ::-1].copy()])
    idx = (np.arange(k) * 7 % 3 == 0).astype(np.int32)
    ent = [5, 6, 7, 8]
    st = ctx.seed_children(ent, 10, k)
    out = ctx.rwalk_batch(prob, u0, axes2, case["scale"], case["loglstar"],
                          walks, st, axes_idx=idx)
    kids = np.random.SeedSequence(ent).spawn(10 + k)[10:]
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        rng = np.random.Generator(bg)
        ref = P.rwalk(u0[i].copy(), case["loglstar"], axes2[idx[i]],
                      case["scale"], prob.prior_transform, prob.loglikelihood,
                      rng, walks)
        assert ref["accept"] == out["accept"][i]
        assert ref["reject"] == out["reject"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=ATOL_U)
        np.testing.assert_allclose(out["logl"][i], ref["logl"], rtol=RTOL_L,
                                   atol=1e-11)
        np.testing.assert_array_equal(out["rng_out"][i],
                                      _lib.pcg_state_words(bg))
    assert out["accept"].sum() > 0 and out["reject"].sum() > 0


def test_problem_eval(ctx):
    for pname in ("C1", "C2", "C3", "G5", "E3", "N6"):
        prob = inputs.problem(pname)
        rng = np.random.default_rng(9)
        u = rng.uniform(0.01, 0.99, size=(200, prob.ndim))
        v, logl = ctx.problem_eval(prob, u)
        np.testing.assert_allclose(v, prob.prior_transform_many(u), rtol=1e-12,
                                   atol=1e-13)
        np.testing.assert_allclose(logl, prob.loglikelihood_many(
            prob.prior_transform_many(u)), rtol=1e-11, atol=1e-11)


def test_rwalk_propose_lockstep(ctx):
    """dh_rwalk_propose (device half of the lock-step path for arbitrary Python
    likelihoods) vs propose_ball_point of the oracle, stream for stream; then a
    full lock-step walk reproduces the fused kernel exactly."""
    from dynesty_amd import _lib
    case = inputs.walker_case("G5", 200, 55, shrink=3.0)
    prob = case["problem"]
    u0 = case["u0"][:100]
    k = len(u0)
    bc = np.zeros(5, dtype=np.int8)
    bc[0] = _lib.BC_PERIODIC
    bc[2] = _lib.BC_REFLECT
    st = ctx.seed_children([77], 0, k)
    up, inside, out = ctx.rwalk_propose(u0, case["axes"], 2.0, st, bc=bc)
    kids = np.random.SeedSequence([77]).spawn(k)
    nonb = bc == 0
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        p, fail = P.propose_ball(u0[i], 2.0, case["axes"], 5,
                                 np.random.Generator(bg), np.array([0]),
                                 np.array([2]), nonb)
        assert inside[i] == (not fail)
        if not fail:
            np.testing.assert_allclose(up[i], p, rtol=0, atol=1e-13)
        np.testing.assert_array_equal(out[i], _lib.pcg_state_words(bg))
    assert inside.any() and (~inside).any()
    # lock-step walk == fused walk
    from dynesty_amd import samplers
    from dynesty_amd.samplers import SamplerReturn  # noqa: F401
    from collections import namedtuple
    Arg = namedtuple("Arg", "u loglstar axes scale prior_transform "
                     "loglikelihood rseed kwargs")
    seeds = np.random.SeedSequence([78]).spawn(16)
    args = [Arg(u0[i].copy(), case["loglstar"], case["axes"], case["scale"],
                prob.prior_transform, prob.loglikelihood, seeds[i],
                dict(walks=20, problem=None, periodic=None, reflective=None))
            for i in range(16)]
    from dynesty_amd import backend
    backend.set_backend(ctx)
    try:
        lock = samplers.run_rwalk(args)
    finally:
        backend.set_backend(None)
    fused = ctx.rwalk_batch(prob, u0[:16], case["axes"], case["scale"],
                            case["loglstar"], 20,
                            ctx.seed_children([78], 0, 16))
    for i in range(16):
        np.testing.assert_allclose(lock[i].u, fused["u"][i], rtol=0, atol=1e-12)
        assert lock[i].tuning_info["accept"] == fused["accept"][i]
