"""RadFriends / SupFriends on the device (friends.hip) against the reference's
golden vectors (tests/golden/friends.npz) and the oracle."""
import os

import numpy as np
import pytest

import inputs
from oracle import friends_ref as F

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "friends.npz"))
CASES = [(k, n) for k in ("balls", "cubes") for n in inputs.CLOUDS_FRIENDS
         if f"{k}/{n}/u1/cov" in GOLD]
# fp64, different summation order / eigensolver than LAPACK: relative to the matrix norm
RTOL = 2e-10


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def close(a, b, rtol=RTOL):
    scale = np.abs(b).max()
    np.testing.assert_allclose(a, b, rtol=0, atol=rtol * scale)


@pytest.mark.parametrize("kind,name", CASES)
def test_update_two_steps(ctx, kind, name):
    """Both updates of the golden sequence, each from the REFERENCE's previous
    metric: the radius puts the loneliest point exactly on the linkage threshold
    of the next update, so whether it becomes its own cluster hangs on the last
    bit of `am` -- given the same `am` the device takes the reference's decision
    (scipy's pdist arithmetic is reproduced), but a chain of device updates
    (Jacobi instead of LAPACK, am equal to ~1e-16) may flip that coin the other
    way, exactly as another LAPACK build would."""
    pts = inputs.cloud(name)
    d = pts.shape[1]
    prev = np.eye(d)
    for step in (1, 2):
        r = ctx.friends_update(pts, kind, am_prev=prev)
        key = f"{kind}/{name}/u{step}"
        for k in ("cov", "am", "axes", "axes_inv"):
            close(r[k], GOLD[f"{key}/{k}"])
        assert abs(r["logvol"] - GOLD[f"{key}/logvol"]) < 1e-9
        prev = GOLD[f"{key}/am"]


@pytest.mark.parametrize("kind,name", CASES)
def test_cluster_count_and_noclustering(ctx, kind, name):
    pts = inputs.cloud(name)
    fr = F.friends_init(kind, pts.shape[1])
    f1, _ = F.friends_update(fr, pts)
    _, info = F.friends_update(f1, pts)
    r = ctx.friends_update(pts, kind, am_prev=f1.am)
    assert r["nclusters"] == info["nclusters"]
    assert abs(r["rmax"] - info["rmax"]) < 1e-11 * info["rmax"] + 1e-14
    r0 = ctx.friends_update(pts, kind, am_prev=None)
    close(r0["cov"], GOLD[f"{kind}/{name}/noclust/cov"])
    assert abs(r0["logvol"] - GOLD[f"{kind}/{name}/noclust/logvol"]) < 1e-9


@pytest.mark.parametrize("kind,name", CASES)
def test_bootstrap_radius(ctx, kind, name):
    pts = inputs.cloud(name)
    n = len(pts)
    fr = F.friends_init(kind, pts.shape[1])
    f1, _ = F.friends_update(fr, pts)
    rstate = np.random.default_rng(9)
    seeds = np.random.SeedSequence(rstate.integers(0, 2**63 - 1, size=4)).spawn(3)
    masks = []
    for s in seeds:
        g = np.random.Generator(np.random.PCG64(s))
        sel = np.zeros(n, dtype=bool)
        sel[np.unique(g.integers(n, size=n))] = True
        masks.append(sel)
    r = ctx.friends_update(pts, kind, am_prev=f1.am, in_masks=np.array(masks))
    key = f"{kind}/{name}/boot"
    for k in ("cov", "am", "axes", "axes_inv"):
        close(r[k], GOLD[f"{key}/{k}"])
    assert abs(r["logvol"] - GOLD[f"{key}/logvol"]) < 1e-9


@pytest.mark.parametrize("kind,name", CASES)
def test_within_exact_indices(ctx, kind, name):
    pts = inputs.cloud(name)
    key = f"{kind}/{name}"
    axes_inv = GOLD[f"{key}/u2/axes_inv"]
    probes = GOLD[f"{key}/probes"]
    counts, bits = ctx.friends_within(pts, kind, axes_inv, probes, want_bits=True)
    np.testing.assert_array_equal(counts, GOLD[f"{key}/within_counts"])
    idx = []
    for row in bits:
        b = np.unpackbits(row.view(np.uint8), bitorder="little")[:len(pts)]
        idx.append(np.nonzero(b)[0])
    got = np.concatenate(idx) if sum(map(len, idx)) else np.zeros(0, np.int64)
    np.testing.assert_array_equal(got, GOLD[f"{key}/within_idx"])
    # and a bulk comparison with the oracle on 500 random candidates
    rng = np.random.default_rng(3)
    fr = F.Friends(kind, GOLD[f"{key}/u2/cov"], GOLD[f"{key}/u2/am"], GOLD[f"{key}/u2/axes"], axes_inv, 0., pts)
    xs = pts[rng.integers(len(pts), size=500)] + 0.5 * rng.standard_normal((500, pts.shape[1])) @ fr.axes
    c2, _ = ctx.friends_within(pts, kind, axes_inv, xs)
    np.testing.assert_array_equal(c2, [len(F.friends_within(fr, x)) for x in xs])


@pytest.mark.parametrize("kind,name", CASES)
def test_draws_same_seed(ctx, kind, name):
    from dynesty_amd import _lib
    pts = inputs.cloud(name)
    key = f"{kind}/{name}"
    axes, axes_inv = GOLD[f"{key}/u2/axes"], GOLD[f"{key}/u2/axes_inv"]
    rs = np.random.default_rng(13)
    xs, qs, out = ctx.friends_draw(_lib.pcg_state6(rs.bit_generator), 12, pts, kind, axes, axes_inv)
    np.testing.assert_allclose(xs, GOLD[f"{key}/samples"], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(out[[0, 1, 4, 5]], GOLD[f"{key}/samples_state_after"])
    rs = np.random.default_rng(14)
    xs, qs, _ = ctx.friends_draw(_lib.pcg_state6(rs.bit_generator), 8, pts, kind, axes, axes_inv, return_q=True)
    np.testing.assert_allclose(xs, GOLD[f"{key}/sample_q_x"], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(qs, GOLD[f"{key}/sample_q_q"])


@pytest.mark.parametrize("kind", ["balls", "cubes"])
def test_unif_friends_batch_same_seed(ctx, kind):
    """Batched UniformBoundSampler inside a friends bound: same streams as the
    oracle (draw order incl. the buffered 32-bit integers and the 1/q uniform),
    same accepted points and call counts."""
    from oracle_backend import OracleBackend
    prob = inputs.problem("C1")
    rng = np.random.default_rng(2)
    live = 0.5 + 0.06 * rng.standard_normal((300, 3))
    fr = F.friends_init(kind, 3)
    fr, _ = F.friends_update(fr, live)
    fr = F.friends_scale_to_logvol(fr, fr.logvol + np.log(1.2))
    _, ll = ctx.problem_eval(prob, live)
    loglstar = float(np.sort(ll)[60])
    states = ctx.seed_children(np.array([5, 6, 7, 8]), 0, 200)
    dev = ctx.unif_friends_batch(prob, loglstar, states, live, kind, fr.axes, fr.axes_inv)
    ref = OracleBackend().unif_friends_batch(prob, loglstar, states, live, kind, fr.axes, fr.axes_inv)
    np.testing.assert_array_equal(dev["ncalls"], ref["ncalls"])
    np.testing.assert_array_equal(dev["rng_out"], ref["rng_out"])
    np.testing.assert_allclose(dev["u"], ref["u"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(dev["logl"], ref["logl"], rtol=1e-11, atol=1e-11)
    assert (dev["logl"] > loglstar).all()


@pytest.mark.parametrize("kind", ["balls", "cubes"])
def test_host_classes_on_device(ctx, kind):
    from dynesty_amd import bounding
    pts = inputs.cloud("egg13")
    b = dict(balls=bounding.HipRadFriends, cubes=bounding.HipSupFriends)[kind](2)
    key = f"{kind}/egg13"
    b.update(pts, rstate=np.random.default_rng(7))
    close(b.cov, GOLD[f"{key}/u1/cov"])
    b.am = GOLD[f"{key}/u1/am"]  # the reference's metric for the knife-edge linkage (see above)
    b.update(pts, rstate=np.random.default_rng(7))
    for k in ("cov", "am", "axes", "axes_inv"):
        close(getattr(b, k), GOLD[f"{key}/u2/{k}"])
    assert abs(b.logvol - GOLD[f"{key}/u2/logvol"]) < 1e-9
    probes = GOLD[f"{key}/probes"]
    assert [b.overlap(x) for x in probes] == list(GOLD[f"{key}/within_counts"])  # one point: host
    assert list(b.overlap_many(probes)) == list(GOLD[f"{key}/within_counts"])  # many points: one launch
    assert b.contains(pts[0]) and not b.contains(np.array([5.0, 5.0]))
    xs = b.samples(12, rstate=np.random.default_rng(13))
    np.testing.assert_allclose(xs, GOLD[f"{key}/samples"], rtol=0, atol=1e-12)
    lv, fu = b.monte_carlo_logvol(2000, rstate=np.random.default_rng(1))
    assert lv < np.log(len(pts)) + b.logvol + 1e-9 and 0.0 < fu <= 1.0


@pytest.mark.parametrize("kind,sample", [("balls", "unif"), ("cubes", "unif"), ("balls", "rwalk"), ("cubes", "rslice")])
def test_end_to_end_c1(ctx, kind, sample):
    """Full static run with bound='balls' / 'cubes' through the dynesty-free driver."""
    from dynesty_amd import nested
    prob = inputs.problem("C1")
    r = nested.run_static(prob, nlive=300, bound=kind, sample=sample, queue_size=32,
                          rstate=np.random.default_rng(17), dlogz=0.1)
    assert abs(r.logz - prob.logz_truth) < 5 * r.logzerr + 0.1
    assert r.nbound > 1
