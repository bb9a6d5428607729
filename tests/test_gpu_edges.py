"""Edge cases and size-independent properties of the device hot path:
determinism, batch == single, largest supported dimensions, degenerate inputs,
odd batch sizes, and the BASELINE full sizes."""
import numpy as np
import pytest

import inputs
from oracle import bounding_ref as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


FIELDS = ("ctrs", "covs", "ams", "axes", "axlens", "logvol_ells")


def test_rebuild_is_deterministic_and_batch_equals_single(ctx):
    """The k-means parts of a node cooperate through a device-scope barrier and
    node ids come from an atomic counter: the RESULT must not depend on any of
    that -- bit-identical across repetitions, and a run inside a 40-run batch
    equals the same run alone."""
    pts = inputs.cloud("c2")
    a = ctx.rebuild(pts, multi=True)
    for _ in range(3):
        b = ctx.rebuild(pts, multi=True)
        assert a["nells"] == b["nells"]
        for k in FIELDS:
            np.testing.assert_array_equal(a[k], b[k])
    rng = np.random.default_rng(0)
    sets = [inputs.cloud("c3")[rng.permutation(5000)[:n]] for n in (5000, 1234, 257, 256, 4999)] * 8
    many = ctx.rebuild_many(sets, multi=True)
    for s, r in list(zip(sets, many))[::7]:
        one = ctx.rebuild(s, multi=True)
        assert one["nells"] == r["nells"]
        for k in FIELDS:
            np.testing.assert_array_equal(one[k], r[k][:one["nells"]] if r[k].shape[0] != one[k].shape[0] else r[k])


@pytest.mark.parametrize("d,n", [(32, 900), (44, 1200), (1, 300), (2, 9)])
def test_dimension_limits_vs_oracle(ctx, d, n):
    """Largest LDS-resident dimensions (32 = last register-resident walker
    dimension, 44 = last LDS rebuild dimension), D = 1, and a node too small to
    split."""
    rng = np.random.default_rng(d)
    a = rng.standard_normal((d, d)) * 0.2 + np.eye(d)
    pts = 0.5 + 0.03 * rng.standard_normal((n, d)) @ a
    if d <= 2:
        pts[: n // 2] += 0.3  # two separated groups
    got = ctx.rebuild(pts, multi=True)
    ref = B.multi_update(pts)
    assert got["nells"] == len(ref.ells)
    order = np.argsort([e.ctr[0] for e in ref.ells])
    mine = np.argsort(got["ctrs"][:, 0])
    for i, j in zip(mine, order):
        e = ref.ells[j]
        np.testing.assert_allclose(got["ctrs"][i], e.ctr, atol=1e-12)
        np.testing.assert_allclose(got["covs"][i], e.cov, rtol=0, atol=1e-8 * np.abs(e.cov).max())
        assert abs(got["logvol_ells"][i] - e.logvol) < 1e-8


def test_degenerate_clouds(ctx):
    """Duplicated points and a cloud confined to a line: improve_covar_mat's
    regularisation path, same ellipsoid volume as the oracle."""
    rng = np.random.default_rng(5)
    base = 0.5 + 0.05 * rng.standard_normal((50, 4))
    dup = np.repeat(base, 6, axis=0)
    line = 0.5 + np.outer(rng.uniform(-0.2, 0.2, 200), np.array([1.0, 2.0, -1.0]) / 3)
    for pts in (dup, line):
        got = ctx.rebuild(pts, multi=False)
        ref = B.bounding_ellipsoid(pts)
        assert abs(got["logvol_ells"][0] - ref.logvol) < 1e-4
        dd = pts - got["ctrs"][0]
        assert np.einsum('ij,jk,ik->i', dd, got["ams"][0], dd).max() < 1.0


@pytest.mark.parametrize("k,walks", [(1, 1), (63, 7), (65, 45), (1000, 3)])
def test_rwalk_odd_batches(ctx, k, walks):
    """Batch sizes that are not multiples of the wavefront, single walker,
    single step: same result as the oracle, walker by walker."""
    from oracle import proposals_ref as P
    case = inputs.walker_case("G5", 64, 3)
    prob = case["problem"]
    u0 = case["u0"][np.arange(k) % len(case["u0"])]
    ent = [9, 9, k, walks]
    states = ctx.seed_children(ent, 0, k)
    out = ctx.rwalk_batch(prob, u0, case["axes"], case["scale"], case["loglstar"], walks, states)
    kids = np.random.SeedSequence(ent).spawn(k)
    for i in list(range(min(k, 8))) + [k - 1]:
        ref = P.rwalk(u0[i].copy(), case["loglstar"], case["axes"], case["scale"], prob.prior_transform,
                      prob.loglikelihood, np.random.Generator(np.random.PCG64(kids[i])), walks)
        assert ref["accept"] == out["accept"][i] and ref["reject"] == out["reject"][i]
        np.testing.assert_allclose(out["u"][i], ref["u"], rtol=0, atol=1e-12)
    assert (out["accept"] + out["reject"] == walks).all()


def test_full_size_properties_c3_and_c5_shard(ctx):
    """BASELINE full sizes through size-independent properties: every live point
    inside the union (strictly), union volume below the single bounding
    ellipsoid, a permutation of the points gives the same set of ellipsoids; 512
    C2 live sets in one launch all succeed."""
    pts = inputs.cloud("c3")
    got = ctx.rebuild(pts, multi=True)
    single = ctx.rebuild(pts, multi=False)
    assert np.logaddexp.reduce(got["logvol_ells"]) < single["logvol_ells"][0]
    count, _, _ = ctx.contains(pts, got["ctrs"], got["ams"], mode=0)
    assert (count >= 1).all()
    perm = np.random.default_rng(1).permutation(len(pts))
    got2 = ctx.rebuild(pts[perm], multi=True)
    assert got2["nells"] == got["nells"]
    np.testing.assert_allclose(np.sort(got2["logvol_ells"]), np.sort(got["logvol_ells"]), atol=1e-7)
    c2 = inputs.cloud("c2")
    rng = np.random.default_rng(2)
    sets = [c2[rng.permutation(2000)] for _ in range(512)]
    many = ctx.rebuild_many(sets, multi=True)
    assert len(many) == 512
    lv = np.array([np.logaddexp.reduce(r["logvol_ells"]) for r in many])
    assert np.ptp(lv) < 1e-6  # the same cloud, permuted: the same bound


@pytest.mark.parametrize("d", [2, 3, 5])
def test_exactly_degenerate_leading_eigenvalues(ctx, d):
    """A lattice cloud has a covariance proportional to the identity: the leading eigenvalues coincide
    exactly, the repeated-squaring iteration of the eigen-free node path cannot converge, and the node must
    take the reference's route in place (regularize with the wave-level Jacobi).  Which of the equal axes is
    'major' is LAPACK's arbitrary choice in the reference, so only properties are checked."""
    g = np.linspace(0.2, 0.8, 7 if d == 2 else 5 if d == 3 else 3)
    pts = np.stack(np.meshgrid(*[g] * d), -1).reshape(-1, d)
    pts = pts[np.random.default_rng(d).permutation(len(pts))]
    a = ctx.rebuild(pts, multi=True, want_labels=True)
    b = ctx.rebuild(pts, multi=True, want_labels=True)
    assert a["nells"] >= 1 and a["nells"] == b["nells"]
    for k in FIELDS:
        np.testing.assert_array_equal(a[k], b[k])
    count, _, _ = ctx.contains(pts, a["ctrs"], a["ams"], mode=0)
    assert (count > 0).all()
    for i in range(a["nells"]):
        ax = a["axes"][i]
        np.testing.assert_allclose(ax @ ax.T, a["covs"][i], rtol=0, atol=1e-10 * np.abs(a["covs"][i]).max())
    # the single-ellipsoid form of the same cloud: the sample covariance itself is c * I
    one = ctx.rebuild(pts, multi=False)
    cov = one["covs"][0]
    np.testing.assert_allclose(cov, np.eye(d) * cov[0, 0], rtol=0, atol=1e-12 * cov[0, 0])
    ref = B.bounding_ellipsoid(pts)
    np.testing.assert_allclose(one["logvol_ells"][0], ref.logvol, rtol=0, atol=1e-9)


def test_cooperative_root_threshold_follows_the_occupancy(ctx):
    """The cooperative root (resident parts meeting at spin barriers) is only used while runs x parts fit the
    chip (capacity from the occupancy API); beyond that the single-workgroup routine runs -- with bit-identical
    results either way."""
    import os
    pts = inputs.cloud("c2")
    sets = [pts[np.random.default_rng(s).permutation(2000)] for s in range(3)]
    coop = [ctx.rebuild(p, multi=True) for p in sets]
    os.environ["DH_ROOT_PARTS"] = "0"
    try:
        single = [ctx.rebuild(p, multi=True) for p in sets]
    finally:
        del os.environ["DH_ROOT_PARTS"]
    for a, b in zip(coop, single):
        assert a["nells"] == b["nells"]
        for k in FIELDS:
            np.testing.assert_array_equal(a[k], b[k])


def test_work_queue_tree_equals_the_level_kernels(ctx):
    """The work-queue form of the tree (DH_TREE=1: persistent workers on one FIFO queue, no levels -- a node's
    k-means is queued when its ellipsoid exists, its children when its last part has partitioned; built in
    round 3, measured slower than the level pipeline and therefore not the default) runs the level kernels'
    node routines, so it must reproduce the level pipeline bit for bit: single live sets and a ragged batch."""
    import os

    def run(fn, tree):
        old = os.environ.get("DH_TREE")
        os.environ["DH_TREE"] = "1" if tree else "0"
        try:
            return fn()
        finally:
            if old is None:
                os.environ.pop("DH_TREE", None)
            else:
                os.environ["DH_TREE"] = old
    many = 0
    for name in ("c2", "c3", "two5", "ring2", "flat10", "small4", "egg13", "c2s"):
        pts = inputs.cloud(name)
        ref = run(lambda: ctx.rebuild(pts, multi=True, want_labels=True), False)
        for rep in range(3):
            got = run(lambda: ctx.rebuild(pts, multi=True, want_labels=True), True)
            assert ref["nells"] == got["nells"] and ref["nnodes"] == got["nnodes"], name
            for k in FIELDS + ("labels",):
                np.testing.assert_array_equal(ref[k], got[k])
        many += ref["nells"] > 4
    assert many >= 2
    # a ragged batch of different live sets in one launch sequence
    sets = [inputs.cloud("c2")[:n] for n in (2000, 1500, 777, 300, 120, 60)] + [inputs.cloud("c2")[::-1].copy()]
    ref = run(lambda: ctx.rebuild_many(sets, multi=True), False)
    got = run(lambda: ctx.rebuild_many(sets, multi=True), True)
    assert len(ref) == len(got) == len(sets)
    for a, b in zip(ref, got):
        assert a["nells"] == b["nells"]
        for k in FIELDS:
            np.testing.assert_array_equal(a[k], b[k])


def test_work_queue_tail_equals_the_level_kernels(ctx):
    """The level kernels are launched for a balanced tree's depth; whatever is deeper is handed to the work-queue
    form (k_tree: persistent workers, any depth, any node size).  Forced to take over early (DH_DEEP_FROM) it must
    give bit-identical results to the level kernels alone (DH_DEEP=0) -- also from level 1 on, where the C2 cloud's
    nodes have 1000 points (the single-workgroup tail of round 2 could only take nodes of <= 256 points)."""
    import os
    cases = [(inputs.cloud("c2"), ("1", "4", "6")), (inputs.cloud("c3"), ("2", "6", "9")), (inputs.cloud("two5"), ("3", "5")),
             (inputs.cloud("ring2"), ("5", "7"))]

    def run(pts, env):
        old = {k: os.environ.get(k) for k in ("DH_DEEP", "DH_DEEP_FROM", "DH_TREE")}
        os.environ.update(dict(env, DH_TREE="0"))  # the level pipeline
        try:
            return ctx.rebuild(pts, multi=True, want_labels=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    deep_trees = 0
    for pts, firsts in cases:
        ref = run(pts, {"DH_DEEP": "0"})
        deep_trees += ref["nells"] > 4
        for env in [{"DH_DEEP_FROM": f} for f in firsts] + [{}]:
            got = run(pts, env)
            assert ref["nells"] == got["nells"] and ref["nnodes"] == got["nnodes"], env
            for k in FIELDS + ("labels",):
                np.testing.assert_array_equal(ref[k], got[k])
    assert deep_trees >= 2


def test_one_wavefront_per_small_node_gives_the_same_bits(ctx):
    """k_ell_wave builds the small nodes of the deep levels with one wavefront each (the same routines instantiated
    for 64 threads, the covariance contraction with k_ell's four waves played in turn): which kernel builds a node is
    a scheduling decision, so every output must be bit-identical with it off (DH_WAVE_ELL=0), on by default
    (D <= 13), leaves only (=1) and with the major axis at any D (=2) -- single live sets and a batch."""
    import os
    clouds = [inputs.cloud(n) for n in ("c3", "two5", "ring2", "g3", "egg13", "c2", "flat10", "small4")]

    def run(pts, env):
        old = os.environ.get("DH_WAVE_ELL")
        if env is None:
            os.environ.pop("DH_WAVE_ELL", None)
        else:
            os.environ["DH_WAVE_ELL"] = env
        try:
            return ctx.rebuild(pts, multi=True, want_labels=True)
        finally:
            if old is None:
                os.environ.pop("DH_WAVE_ELL", None)
            else:
                os.environ["DH_WAVE_ELL"] = old
    many = 0
    for pts in clouds:
        ref = run(pts, "0")
        many += ref["nells"] > 4
        for env in (None, "1", "2"):
            got = run(pts, env)
            assert ref["nells"] == got["nells"] and ref["nnodes"] == got["nnodes"], env
            for k in FIELDS + ("labels",):
                np.testing.assert_array_equal(ref[k], got[k])
    assert many >= 3
    # a batch of permuted eggbox live sets (the rounds of workgroup slots are what the wave form is for)
    c3 = inputs.cloud("c3")
    sets = [c3[np.random.default_rng(r).permutation(len(c3))] for r in range(8)]
    os.environ["DH_WAVE_ELL"] = "0"
    try:
        ref = ctx.rebuild_many(sets, multi=True)
    finally:
        del os.environ["DH_WAVE_ELL"]
    got = ctx.rebuild_many(sets, multi=True)
    for a, b in zip(ref, got):
        assert a["nells"] == b["nells"]
        for k in a:
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))


def test_unbalanced_tree_deeper_than_the_level_plan(ctx):
    """A chain of unbalanced splits -- every 2-means split peels one small far cluster off a big rest -- is deeper
    than the levels the launch plan gives a balanced tree, with nodes of far more than a tile of points down there
    (the serial tail failed such a run with DH_ERR_NOMEM).  The work-queue tail builds it; all level
    kernels (DH_DEEP=0: 2 log2(n / 2d) + 8 of them) give the same bits."""
    import os
    rng = np.random.default_rng(5)
    parts = [0.5 + 0.0005 * rng.standard_normal((900, 2))]
    for k in range(14):  # clusters further and further out, each far from everything nearer
        c = np.array([0.5 + 0.45 * 0.62 ** k, 0.5 + 0.3 * 0.62 ** k])
        parts.append(c + 0.02 * 0.62 ** k * 0.02 * rng.standard_normal((70, 2)))
    pts = np.concatenate(parts)[rng.permutation(900 + 14 * 70)]
    got = ctx.rebuild(pts, multi=True, want_labels=True)
    os.environ["DH_DEEP"] = "0"
    try:
        ref = ctx.rebuild(pts, multi=True, want_labels=True)
    finally:
        del os.environ["DH_DEEP"]
    assert got["nells"] == ref["nells"] and got["nnodes"] == ref["nnodes"] and got["nells"] >= 10
    for k in FIELDS + ("labels",):
        np.testing.assert_array_equal(ref[k], got[k])
    # the oracle's tree on the same cloud: same number of ellipsoids, same partition of the points
    from oracle import bounding_ref as B
    m = B.multi_update(pts)
    assert m.nells == got["nells"]


def test_leaves_on_the_side_stream_give_the_same_bits(ctx):
    """Round 5: above D = 13 the leaves of the deep levels (count < 4 d: read by nothing but the accept test) are
    built by k_ell_wave<128> on the side stream beside the level kernels, and a leaf its eigen-free path declines is
    queued for the work-queue tail.  Which kernel builds a node is a scheduling decision: every output bit-identical
    with it off (DH_LEAF_SIDE=0) -- blobs whose leaves are healthy, blobs whose leaves are DEGENERATE (duplicated
    points, clusters confined to a plane: the declined route), single live sets and a batch."""
    import os
    rng = np.random.default_rng(11)
    d = 16

    def blobs(nblob, per, kind):
        parts = []
        for b in range(nblob):
            c = rng.uniform(0.2, 0.8, d)
            if kind == "healthy":
                x = c + 0.01 * rng.standard_normal((per, d))
            elif kind == "dup":  # per / 3 distinct points, three copies each
                x = np.repeat(c + 0.01 * rng.standard_normal((per // 3, d)), 3, axis=0)
            else:  # confined to a 5-dimensional plane
                x = c + 0.01 * rng.standard_normal((per, 5)) @ rng.standard_normal((5, d))
            parts.append(x)
        pts = np.concatenate(parts)
        return pts[rng.permutation(len(pts))]
    # 16 blobs of ~40 points at d = 16: the leaves (count < 4 d = 64) sit at level 3, the first level whose average
    # child is below 3 d points, i.e. the level the side-stream kernel takes
    clouds = [blobs(16, 40, "healthy"), blobs(16, 39, "dup"), blobs(16, 40, "plane"), inputs.cloud("c2")]

    def run(pts, off):
        old = os.environ.get("DH_LEAF_SIDE")
        os.environ["DH_LEAF_SIDE"] = "0" if off else "1"  # (round 6: the side stream is opt-in)
        try:
            return ctx.rebuild(pts, multi=True, want_labels=True)
        finally:
            if old is None:
                os.environ.pop("DH_LEAF_SIDE", None)
            else:
                os.environ["DH_LEAF_SIDE"] = old
    for pts in clouds:
        ref, got = run(pts, True), run(pts, False)
        assert ref["nells"] == got["nells"] and ref["nnodes"] == got["nnodes"]
        for k in FIELDS + ("labels",):
            np.testing.assert_array_equal(ref[k], got[k])
    assert run(clouds[0], False)["nells"] >= 5
    # a batch: 16 permutations of the degenerate cloud (declined leaves of many runs in one queue)
    sets = [clouds[1][np.random.default_rng(r).permutation(len(clouds[1]))] for r in range(16)]
    os.environ["DH_LEAF_SIDE"] = "0"
    try:
        ref = ctx.rebuild_many(sets, multi=True)
        os.environ["DH_LEAF_SIDE"] = "1"
        got = ctx.rebuild_many(sets, multi=True)
    finally:
        del os.environ["DH_LEAF_SIDE"]
    for a, b in zip(ref, got):
        assert a["nells"] == b["nells"]
        for k in a:
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))


def test_root_of_more_runs_than_fit_at_once_goes_in_chunks_with_the_same_bits(ctx):
    """Round 6: the cooperative root (k_root_parts: ceil(n / 256) workgroups per run that meet at spin waits) needs
    its workgroups co-resident; 144 runs x 8 parts do not fit the 512 slots, so the runs go in chunks of 64 -- same
    bits as the single-workgroup root the launcher fell back to before (DH_ROOT_CHUNK=0) and as 144 separate calls'
    first runs."""
    import os
    base = inputs.cloud("c2")
    sets = [base[np.random.default_rng(r).permutation(len(base))] for r in range(144)]
    got = ctx.rebuild_many(sets, multi=True)
    os.environ["DH_ROOT_CHUNK"] = "0"
    try:
        ref = ctx.rebuild_many(sets, multi=True)
    finally:
        del os.environ["DH_ROOT_CHUNK"]
    assert len(got) == len(ref) == 144
    for a, b in zip(ref, got):
        assert a["nells"] == b["nells"] and a["nells"] >= 1
        for k in a:
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))
    for r in (0, 63, 64, 143):  # either side of a chunk boundary against the run on its own
        one = ctx.rebuild(sets[r], multi=True)
        for k in FIELDS:
            np.testing.assert_array_equal(np.asarray(one[k]), np.asarray(got[r][k]))


def test_capped_level_grids_give_the_same_bits(ctx):
    """Round 6: k_ell / k_ell_wave are launched with at most a few rounds of the chip's workgroup slots and loop over
    their run's children (DH_LEVEL_GRID_CAP=0: the worst-case grids).  A many-mode cloud in two dimensions -- hundreds of
    small nodes per level, the shape the cap is for -- in a batch large enough for the cap to bind: every output the same."""
    import os
    rng = np.random.default_rng(21)
    ctrs = rng.uniform(0.1, 0.9, (40, 2))
    base = np.concatenate([c + 0.004 * rng.standard_normal((60, 2)) for c in ctrs])
    sets = [base[np.random.default_rng(r).permutation(len(base))] for r in range(48)]
    got = ctx.rebuild_many(sets, multi=True)
    os.environ["DH_LEVEL_GRID_CAP"] = "0"
    try:
        ref = ctx.rebuild_many(sets, multi=True)
    finally:
        del os.environ["DH_LEVEL_GRID_CAP"]
    assert max(g["nells"] for g in got) >= 10
    for a, b in zip(ref, got):
        assert a["nells"] == b["nells"]
        for k in a:
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))
