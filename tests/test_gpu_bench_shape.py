"""The exact launch shape bench.py times -- dh_rebuild_batch_dev + dh_enlarge_batch_dev +
dh_rwalk_batch_dev over 64 runs x 2000 live points x 25 dims with one frame per run
(`axes_idx` = run * MAX_ELLS, 64 x 8 cooperative root parts) -- held to the oracle:
runs 0 / 31 / 63 against oracle.multi_update + scale_multi_to_logvol, and 64 walkers of each of
those runs against oracle.rwalk on the same SeedSequence child streams (bench.Shard.verify, which
bench.py also runs after its timed region)."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def test_bench_shape_vs_oracle(ctx):
    sh = bench.Shard(ctx, bench.c2_problem(), runs=64, nlive=2000, walks=45, seed=1000,
                     entropy=(21, 0, 0, 0))
    # a few timed-style steps first (ping-pong generator buffers), as the benchmark does
    sh.rebuild()
    for i in range(3):
        sh.step(i)
    ctx.sync()
    info = sh.verify(check_runs=[0, 31, 63], walkers_per_run=64)
    assert info["ok"] and info["walkers_checked"] == 240
    # every run of the shard: one ellipsoid for a unimodal contour, all live points inside
    b = sh.fetch_bound()
    assert np.all(b["status"] == 0)
    for r in range(sh.runs):
        pts = sh.u0[r * sh.nlive:(r + 1) * sh.nlive]
        inside = np.zeros(len(pts), bool)
        for j in range(int(b["nells"][r])):
            dlt = pts - b["ctrs"][r, j]
            inside |= np.einsum('ij,jk,ik->i', dlt, b["ams"][r, j], dlt) < 1.0
        assert inside.all()


def test_bench_shape_batch_equals_single(ctx):
    """A run inside the 64-run batch == the same live set rebuilt alone through the host entry
    point (dh_rebuild + dh_scale_to_logvol), bit for bit."""
    sh = bench.Shard(ctx, bench.c2_problem(), runs=64, nlive=2000, walks=45, seed=1000)
    sh.rebuild(enlarge=False)
    ctx.sync()
    b = sh.fetch_bound()
    for r in (0, 17, 63):
        one = ctx.rebuild(sh.u0[r * sh.nlive:(r + 1) * sh.nlive], multi=True, max_ells=bench.MAX_ELLS)
        m = one["nells"]
        assert m == int(b["nells"][r])
        np.testing.assert_array_equal(one["ctrs"], b["ctrs"][r, :m])
        np.testing.assert_array_equal(one["covs"], b["covs"][r, :m])
        np.testing.assert_array_equal(one["axes"], b["axes"][r, :m])
        np.testing.assert_array_equal(one["logvol_ells"], b["logvols"][r, :m])


def test_many_runs_in_chunks_equal_single(ctx):
    """160 runs in one batch: k_split's workgroups no longer fit the chip together, so its grid goes in chunks of
    runs (part-major inside a chunk, chunks in index order: the parts of a node meet at spin barriers), and the top
    levels' k_ell workgroups are more than one per CU, so they stage 256 points at a time where the 64-run batch
    stages 512.  Every run checked == the same live set rebuilt alone, bit for bit."""
    sh = bench.Shard(ctx, bench.c2_problem(), runs=160, nlive=2000, walks=45, seed=1000)
    sh.rebuild(enlarge=False)
    ctx.sync()
    b = sh.fetch_bound()
    assert np.all(b["status"] == 0)
    for r in (0, 63, 64, 101, 159):
        one = ctx.rebuild(sh.u0[r * sh.nlive:(r + 1) * sh.nlive], multi=True, max_ells=bench.MAX_ELLS)
        m = one["nells"]
        assert m == int(b["nells"][r])
        np.testing.assert_array_equal(one["ctrs"], b["ctrs"][r, :m])
        np.testing.assert_array_equal(one["covs"], b["covs"][r, :m])
        np.testing.assert_array_equal(one["axes"], b["axes"][r, :m])
        np.testing.assert_array_equal(one["logvol_ells"], b["logvols"][r, :m])


def test_small_shard_other_seeds(ctx):
    """Odd sizes through the same entry points (runs not a multiple of anything, one run)."""
    for runs, seed in ((1, 5), (3, 6)):
        sh = bench.Shard(ctx, bench.c2_problem(), runs=runs, nlive=2000, walks=45, seed=seed,
                         entropy=(7, seed))
        info = sh.verify(check_runs=list(range(runs)), walkers_per_run=16)
        assert info["ok"]
