"""rwalk with four lanes per walker (csrc/walkq.hip: frame product and Gaussian quadratic form on the fp64
matrix cores, PCG64 consumed by four lanes through LCG jumps) against the oracle's restatement of
RWalkSampler.sample (internal_samplers.py:866-1035) on the same child streams -- accept / reject counts and
the generator end state exact, u within 1e-12, logl 1e-11 relative -- over every padded register count the
kernel is built for, several frames inside one wavefront, ragged batch sizes and the three fused
likelihoods; and against the lane-per-walker kernel (csrc/walk.hip), which consumes the same streams."""
import math

import numpy as np
import pytest

from dynesty_amd import problems
from oracle import proposals_ref as P

pytestmark = pytest.mark.gpu

ATOL_U = 1e-12
RTOL_L = 1e-11


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    c = _lib.Context(0)
    yield c
    c.set_rwalk_form(0)
    c.set_rwalk_items(True, 1 << 30)


def make_case(prob, nwalk, seed, shrink=0.5):
    """tests/inputs.walker_case for an arbitrary problem object."""
    d = prob.ndim
    rng = np.random.default_rng(seed)
    if prob.prior_id == problems.PRIOR_NORMAL:
        u0 = 0.5 + 0.08 * rng.standard_normal((nwalk, d))
        spread = 0.08
    elif prob.like_id == problems.LIKE_EGGBOX:
        u0 = 0.5 + 0.004 * rng.standard_normal((nwalk, d))
        spread = 0.004
    else:
        hw = prob.prior_par[0]
        u0 = 0.5 + (shrink / (2 * hw)) * rng.standard_normal((nwalk, d))
        spread = shrink / (2 * hw)
    u0 = np.clip(u0, 1e-3, 1 - 1e-3)
    logl = prob.loglikelihood_many(prob.prior_transform_many(u0))
    loglstar = float(np.quantile(logl, 0.05))
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    axes = q * (spread * math.sqrt(d) * rng.uniform(0.8, 1.6, size=d))
    keep = logl > loglstar
    return dict(u0=u0[keep], loglstar=loglstar, axes=axes, scale=0.7, problem=prob)


def the_problem(kind, d):
    if kind == "prec":
        return problems.gauss_corr(d, 0.4, 5.0, f"corr{d}")
    if kind == "iid":
        return problems.gauss_iid(d, 6.0, f"iid{d}")
    if kind == "normal":  # Normal prior (ndtri), iid Normal likelihood: the C4 family (round 4: built here too)
        return problems.gauss_normal_prior(d, f"nprior{d}")
    return problems.eggbox(d, name=f"egg{d}")


CASES = [("prec", 2), ("prec", 4), ("prec", 5), ("prec", 8), ("iid", 3), ("iid", 7), ("egg", 2), ("egg", 6),
         ("normal", 4), ("normal", 8), ("prec", 9), ("prec", 12), ("prec", 16), ("prec", 17), ("prec", 20), ("prec", 25), ("prec", 28),
         ("prec", 29), ("prec", 32), ("iid", 10), ("iid", 27), ("egg", 9), ("egg", 18), ("normal", 11), ("normal", 25), ("normal", 30)]


@pytest.mark.parametrize("kind,d", CASES)
def test_quad_rwalk_vs_oracle(ctx, kind, d):
    """Three frames mixed inside every wavefront (16 walkers per wave), a batch size off every
    granularity of the kernel (16 walkers per wave, 64 per workgroup)."""
    from dynesty_amd import _lib
    ctx.set_rwalk_form(2)
    prob = the_problem(kind, d)
    case = make_case(prob, 120, 1000 + d)
    walks = 18
    u0 = case["u0"][:83]
    k = len(u0)
    a = case["axes"]
    axes3 = np.stack([a, 0.5 * a[::-1, ::-1].copy(), 0.8 * a.T.copy()])
    idx = (np.arange(k) * 5 % 3).astype(np.int32)
    ent = [d, 6, 7]
    st = ctx.seed_children(ent, 3, k)
    out = ctx.rwalk_batch(prob, u0, axes3, case["scale"], case["loglstar"], walks, st, axes_idx=idx)
    kids = np.random.SeedSequence(ent).spawn(3 + k)[3:]
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        ref = P.rwalk(u0[i].copy(), case["loglstar"], axes3[idx[i]], case["scale"], prob.prior_transform,
                      prob.loglikelihood, np.random.Generator(bg), walks)
        assert ref["accept"] == out["accept"][i], (i, ref["accept"], out["accept"][i])
        assert ref["reject"] == out["reject"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=ATOL_U)
        np.testing.assert_allclose(out["v"][i], prob.prior_transform(ref["u"]), rtol=0, atol=2e-11)
        np.testing.assert_allclose(out["logl"][i], ref["logl"], rtol=RTOL_L, atol=1e-11)
        np.testing.assert_array_equal(out["rng_out"][i], _lib.pcg_state_words(bg))
    assert out["accept"].sum() > 0 and out["reject"].sum() > 0


def test_quad_equals_lane_form_and_batch_halves(ctx):
    """Both kernel forms walk the same streams: counts and generator end states identical, coordinates to
    rounding.  And a walker's result does not depend on its company: a batch equals its halves bit for bit
    (split off the wave / workgroup granularity)."""
    prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
    case = make_case(prob, 6000, 5)
    u0 = case["u0"][:5003]
    k = len(u0)
    st = ctx.seed_children([11, 12], 0, k)
    args = (prob, u0, case["axes"], case["scale"], case["loglstar"], 45, st)
    ctx.set_rwalk_form(1)
    lane = ctx.rwalk_batch(*args)
    ctx.set_rwalk_form(2)
    quad = ctx.rwalk_batch(*args)
    np.testing.assert_array_equal(quad["accept"], lane["accept"])
    np.testing.assert_array_equal(quad["reject"], lane["reject"])
    np.testing.assert_array_equal(quad["rng_out"], lane["rng_out"])
    np.testing.assert_allclose(quad["u"], lane["u"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(quad["logl"], lane["logl"], rtol=1e-12, atol=1e-12)
    cut = 2501
    h0 = ctx.rwalk_batch(prob, u0[:cut], case["axes"], case["scale"], case["loglstar"], 45, st[:cut])
    h1 = ctx.rwalk_batch(prob, u0[cut:], case["axes"], case["scale"], case["loglstar"], 45, st[cut:])
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(quad[key], np.concatenate([h0[key], h1[key]]))


def test_generator_pass_equals_fused_generator_and_chunks(ctx):
    """Round 4: the walkers' PCG64 item streams come from a generator pass ahead of the walk (itemgen_kernel) or from
    the generator inside the walk kernel -- the same rounds on the same streams: every output bit for bit, also
    when the pass's buffer is so small that the launch goes in chunks of walkers (chunk edges off every
    granularity), with several frames and ragged sizes."""
    for d, k, walks in ((25, 3001, 45), (12, 517, 20), (32, 260, 7)):
        prob = problems.gauss_corr(d, 0.4, 5.0, f"corr{d}")
        case = make_case(prob, k + 400, 40 + d)
        u0 = case["u0"][:k]
        assert len(u0) == k
        a = case["axes"]
        axes3 = np.stack([a, 0.5 * a[::-1, ::-1].copy(), 0.8 * a.T.copy()])
        idx = (np.arange(k) * 7 % 3).astype(np.int32)
        st = ctx.seed_children([d, 1, 2], 5, k)
        args = (prob, u0, axes3, case["scale"], case["loglstar"], walks, st)
        ctx.set_rwalk_form(2)
        ctx.set_rwalk_items(False)
        fused = ctx.rwalk_batch(*args, axes_idx=idx)
        ctx.set_rwalk_items(True, 1 << 30)
        one = ctx.rwalk_batch(*args, axes_idx=idx)
        ctx.set_rwalk_items(True, 200 * walks * (d + 1) * 8)  # 200 walkers' streams -> chunks of 192
        chunks = ctx.rwalk_batch(*args, axes_idx=idx)
        ctx.set_rwalk_items(True, 1 << 30)
        for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
            np.testing.assert_array_equal(one[key], fused[key], err_msg=key)
            np.testing.assert_array_equal(chunks[key], fused[key], err_msg=key)
        assert fused["accept"].sum() > 0 and fused["reject"].sum() > 0


def test_a_walker_does_not_depend_on_the_launch_size(ctx):
    """the kernel form used to follow the launch size (four lanes per walker up to 256 x #CU
    walkers, one per lane above), and the two agree only to rounding, so a run's accept / reject sequence could
    depend on how many runs share a GPU.  The form is a function of the problem alone now: walkers of a launch on
    the far side of the old threshold equal the same walkers launched alone, bit for bit."""
    prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
    case = make_case(prob, 9000, 77)
    base = case["u0"][:8192]
    reps = 10  # 81 920 walkers > 256 x 256
    u0 = np.tile(base, (reps, 1))
    k = len(u0)
    st = ctx.seed_children([3, 1, 4], 0, k)
    ctx.set_rwalk_form(0)
    big = ctx.rwalk_batch(prob, u0, case["axes"], case["scale"], case["loglstar"], 12, st)
    lo, hi = 70001, 70001 + 333
    small = ctx.rwalk_batch(prob, u0[lo:hi], case["axes"], case["scale"], case["loglstar"], 12, st[lo:hi])
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(big[key][lo:hi], small[key], err_msg=key)


def test_quad_philox_equals_lane_philox(ctx):
    """Throughput RNG mode: the four-lane kernel positions hiprand's Philox stream exactly where the
    lane-per-walker kernel reads it (per step ceil(n / 4) normal4 blocks, one uniform double)."""
    prob = problems.gauss_corr(25, 0.4, 5.0, "C2")
    case = make_case(prob, 3000, 6)
    u0 = case["u0"][:2049]
    args = (prob, u0, case["axes"], case["scale"], case["loglstar"], 45, 1234)
    ctx.set_rwalk_form(1)
    lane = ctx.rwalk_batch_philox(*args, sequence0=17, offset=4096)
    ctx.set_rwalk_form(2)
    quad = ctx.rwalk_batch_philox(*args, sequence0=17, offset=4096)
    assert (quad["accept"] != lane["accept"]).mean() < 1e-3  # same draws; a knife-edge accept may differ
    same = quad["accept"] == lane["accept"]
    np.testing.assert_allclose(quad["u"][same], lane["u"][same], rtol=0, atol=1e-12)
    assert 0.05 < quad["accept"].mean() / 45 < 0.8


@pytest.mark.parametrize("d", [5, 8, 9, 14, 25, 32])
def test_quad_periodic_and_reflective_coordinates(ctx, d):
    """Round 4: the four-lanes-per-walker kernel wraps periodic and reflects reflective coordinates itself
    (utils.py:1036-1078), so a problem with such coordinates takes the same form as one without.  Large steps, so
    that wraps, reflections and hard-edge rejects all occur; against the oracle on the same streams (counts and
    generator end states exact), against the lane-per-walker kernel, for both generator placements and for the
    Philox mode."""
    from dynesty_amd import _lib
    prob = problems.gauss_corr(d, 0.4, 5.0, f"corr{d}")
    case = make_case(prob, 200, 77 + d)
    u0 = case["u0"][:131]
    k = len(u0)
    bc = np.zeros(d, dtype=np.int8)
    bc[[0, 3, d - 1]] = _lib.BC_PERIODIC
    bc[[1, d - 2]] = _lib.BC_REFLECT
    periodic, reflective = np.where(bc == 1)[0], np.where(bc == 2)[0]
    nonb = bc == 0
    axes = case["axes"] * 2.0          # about half of the proposals leave the cube in some hard coordinate
    loglstar = -1e300                  # ... and every one that stays is accepted: the wrapped point is what is kept
    ent = [d, 60, 7]
    st = ctx.seed_children(ent, 0, k)
    args = (prob, u0, axes, 1.0, loglstar, 20, st)
    ctx.set_rwalk_form(2)
    quad = ctx.rwalk_batch(*args, bc=bc)
    kids = np.random.SeedSequence(ent).spawn(k)
    wrapped = 0
    for i in range(k):
        bg = np.random.PCG64(kids[i])
        ref = P.rwalk(u0[i].copy(), loglstar, axes, 1.0, prob.prior_transform, prob.loglikelihood,
                      np.random.Generator(bg), 20, periodic=periodic, reflective=reflective, nonbounded=nonb)
        assert ref["accept"] == quad["accept"][i] and ref["reject"] == quad["reject"][i], i
        np.testing.assert_allclose(quad["u"][i], ref["u"], rtol=0, atol=ATOL_U)
        np.testing.assert_allclose(quad["logl"][i], ref["logl"], rtol=RTOL_L, atol=1e-11)
        np.testing.assert_array_equal(quad["rng_out"][i], _lib.pcg_state_words(bg))
        wrapped += ref["accept"]
    assert wrapped > k and quad["reject"].sum() > k   # both outcomes, many times
    assert (quad["u"] > 0).all() and (quad["u"] < 1).all()
    hard = ctx.rwalk_batch(*args)
    assert (hard["accept"] < quad["accept"]).any()   # proposals that only survive wrapped / reflected
    ctx.set_rwalk_items(False, 0)
    fused = ctx.rwalk_batch(*args, bc=bc)
    ctx.set_rwalk_items(True, 1 << 30)
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(fused[key], quad[key])
    ctx.set_rwalk_form(1)
    lane = ctx.rwalk_batch(*args, bc=bc)
    np.testing.assert_array_equal(quad["accept"], lane["accept"])
    np.testing.assert_array_equal(quad["rng_out"], lane["rng_out"])
    np.testing.assert_allclose(quad["u"], lane["u"], rtol=0, atol=1e-13)
    lane_ph = ctx.rwalk_batch_philox(prob, u0, axes, 1.0, loglstar, 20, 99, bc=bc)
    ctx.set_rwalk_form(2)
    quad_ph = ctx.rwalk_batch_philox(prob, u0, axes, 1.0, loglstar, 20, 99, bc=bc)
    np.testing.assert_array_equal(quad_ph["accept"], lane_ph["accept"])
    np.testing.assert_allclose(quad_ph["u"], lane_ph["u"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("d", [5, 9, 13, 17, 21, 25, 29])
def test_last_column_by_vector_instructions_equals_the_matrix_form(ctx, d):
    """Round 6: at n = 4 (NR - 1) + 1 the last K step of the frame product and of the Gaussian quadratic form holds one
    live column; the R1 instance of rwalkq_kernel adds it as fma(column, x, acc) by vector instructions instead of a
    matrix instruction per row block (DH_RWALKQ_R1=0 keeps the matrix form).  The matrix instruction's K steps are fused
    multiply-adds and the padded products exact zeros, so every bit must agree: u, v, ln L, counts, generator states --
    PCG64 items and Philox rows, one frame and mixed frames, a batch off every granularity."""
    import os
    prob = problems.gauss_corr(d, 0.4, 5.0, f"corr{d}")
    case = make_case(prob, 5200, 700 + d)
    u0 = case["u0"][:4099]
    k = len(u0)
    a = case["axes"]
    axes3 = np.stack([a, 0.5 * a[::-1, ::-1].copy(), 0.8 * a.T.copy()])
    idx = ((np.arange(k) // 40) % 3).astype(np.int32)
    st = ctx.seed_children([d, 3], 1, k)
    ctx.set_rwalk_form(2)
    ctx.set_rwalk_items(True, 1 << 30)

    def both(fn):
        out = []
        for v in ("1", "0"):
            os.environ["DH_RWALKQ_R1"] = v
            try:
                out.append(fn())
            finally:
                del os.environ["DH_RWALKQ_R1"]
        return out
    r1, mat = both(lambda: ctx.rwalk_batch(prob, u0, a, case["scale"], case["loglstar"], 45, st))
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(r1[key], mat[key], err_msg=key)
    assert r1["accept"].sum() > 0 and r1["reject"].sum() > 0
    r1, mat = both(lambda: ctx.rwalk_batch(prob, u0, axes3, case["scale"], case["loglstar"], 30, st, axes_idx=idx))
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(r1[key], mat[key], err_msg=key)
    r1, mat = both(lambda: ctx.rwalk_batch_philox(prob, u0, a, case["scale"], case["loglstar"], 45, 4321, sequence0=3))
    for key in ("u", "v", "logl", "accept", "reject"):
        np.testing.assert_array_equal(r1[key], mat[key], err_msg=key)
    # periodic and reflective coordinates beside hard ones (a wide proposal so that wraps happen)
    bc = (np.arange(d) % 3).astype(np.int8)
    r1, mat = both(lambda: ctx.rwalk_batch(prob, u0, 6.0 * a, case["scale"], case["loglstar"] - 50.0, 20, st, bc=bc))
    for key in ("u", "v", "logl", "accept", "reject", "rng_out"):
        np.testing.assert_array_equal(r1[key], mat[key], err_msg=key)
    assert r1["accept"].sum() > 0
