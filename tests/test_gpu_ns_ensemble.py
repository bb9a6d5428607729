"""Device-resident ensemble of nested-sampling runs (dh_ns_ensemble, SURVEY.md
8f-1 / BASELINE config C5): statistics vs the analytic evidence, determinism,
and independence of the way the ensemble is sharded."""
import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


@pytest.mark.parametrize("pname,bound,nlive,K,walks", [
    ("G5", "multi", 400, 64, 25), ("C1", "single", 400, 64, 23),
    ("C3", "multi", 600, 128, 22)])
def test_ensemble_logz(ctx, pname, bound, nlive, K, walks):
    prob = inputs.problem(pname)
    r = ctx.ns_ensemble(prob, 16, nlive, K, walks=walks, bound=bound,
                        entropy=[7, 7], dlogz=0.05)
    assert np.all(r["status"] == 0)
    assert np.all(r["nbound"] >= 2)
    lz = r["logz"]
    se = lz.std(ddof=1) / np.sqrt(len(lz))
    # ensemble mean within 5 standard errors (+ the dlogz truncation) of the truth
    assert abs(lz.mean() - prob.logz_truth) < 5 * se + 0.08, (lz.mean(), se)
    # the per-run error estimate sqrt(H/N) describes the scatter
    assert 0.4 < lz.std(ddof=1) / r["logzerr"].mean() < 2.5


@pytest.mark.parametrize("pname,sample,slices,nlive,K", [
    ("G5", "rslice", 5, 400, 64), ("G5", "slice", 3, 400, 64),
    ("C3", "rslice", 5, 1000, 256)])
def test_ensemble_slice_samplers(ctx, pname, sample, slices, nlive, K):
    """BASELINE config C3 shape (eggbox, multi/rslice) and the slice samplers in
    the device-resident loop (tune_slice, per-run doubling flag)."""
    prob = inputs.problem(pname)
    r = ctx.ns_ensemble(prob, 16, nlive, K, bound="multi", sample=sample,
                        slices=slices, entropy=[11], dlogz=0.05)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    se = lz.std(ddof=1) / np.sqrt(len(lz))
    assert abs(lz.mean() - prob.logz_truth) < 5 * se + 0.08, (lz.mean(), se)


def test_deterministic_and_sharding_independent(ctx):
    prob = inputs.problem("G5")
    kw = dict(nlive=200, queue_size=64, walks=20, bound="multi",
              entropy=[3, 1, 4], dlogz=0.5, rebuild_sync=False)
    a = ctx.ns_ensemble(prob, 6, **kw)
    b = ctx.ns_ensemble(prob, 6, **kw)
    np.testing.assert_array_equal(a["logz"], b["logz"])
    np.testing.assert_array_equal(a["ncall"], b["ncall"])
    lo = ctx.ns_ensemble(prob, 3, first_run=0, **kw)
    hi = ctx.ns_ensemble(prob, 3, first_run=3, **kw)
    np.testing.assert_array_equal(np.concatenate([lo["logz"], hi["logz"]]),
                                  a["logz"])
    np.testing.assert_array_equal(np.concatenate([lo["niter"], hi["niter"]]),
                                  a["niter"])


def test_dead_points_ordered(ctx):
    prob = inputs.problem("C1")
    r = ctx.ns_ensemble(prob, 2, 200, 32, walks=23, bound="single",
                        entropy=[5], dlogz=0.5, max_iter=20000,
                        want_dead_logl=True)
    for i in range(2):
        n = int(r["niter"][i])
        d = r["dead_logl"][i, :n]
        assert np.all(np.diff(d) >= 0)  # worst-first: non-decreasing


def test_merge_runs_logz(ctx):
    """Ensemble combiner (the evidence part of utils.merge_runs): 16 runs of 200
    live points merged into one 3200-point run."""
    from dynesty_amd import ensemble
    prob = inputs.problem("G5")
    r = ctx.ns_ensemble(prob, 16, 200, 64, walks=25, bound="multi",
                        entropy=[13], dlogz=0.01, max_iter=50000,
                        want_dead_logl=True)
    lz, err = ensemble.merge_logz(r["dead_logl"], r["niter"], r["live_logl"])
    # merged estimate: much tighter than a single run, consistent with the mean
    assert err < 0.6 * r["logzerr"].mean()
    assert abs(lz - prob.logz_truth) < 5 * err + 0.1
    assert abs(lz - r["logz"].mean()) < 0.15


def test_rebuild_sync_mode(ctx):
    """rebuild_sync=True: all runs rebuild together (early, never late).  Same
    statistics, at most as many fills, more bound updates per run than the
    per-run schedule."""
    prob = inputs.problem("G5")
    kw = dict(nlive=300, queue_size=64, walks=25, bound="multi", entropy=[8],
              dlogz=0.1)
    ref = ctx.ns_ensemble(prob, 16, rebuild_sync=False, **kw)
    syn = ctx.ns_ensemble(prob, 16, rebuild_sync=True, **kw)
    assert (syn["status"] == 0).all() and (ref["status"] == 0).all()
    assert syn["nbound"].mean() >= ref["nbound"].mean()
    for r in (ref, syn):
        se = r["logz"].std(ddof=1) / 4.0
        assert abs(r["logz"].mean() - prob.logz_truth) < 5 * se + 0.05
    # deterministic for a fixed ensemble
    again = ctx.ns_ensemble(prob, 16, rebuild_sync=True, **kw)
    np.testing.assert_array_equal(again["logz"], syn["logz"])


def test_samples_and_merged_run(ctx):
    """want_samples: the stored dead / live coordinates reproduce the stored
    log-likelihoods, and the merged run (ensemble.merge_static_runs) gives the
    analytic posterior moments and ln Z of the Gaussian test problem."""
    from dynesty_amd import ensemble
    prob = inputs.problem("G5")
    m = ensemble.run_ensemble_merged(prob, 8, nlive=300, queue_size=64,
                                     entropy=[9], walks=25, bound="multi",
                                     dlogz=0.1, max_iter=20000)
    r = m["runs"]
    for run in range(8):
        k = int(r["niter"][run])
        _, ll = ctx.problem_eval(prob, r["dead_u"][run, :k])
        np.testing.assert_allclose(ll, r["dead_logl"][run, :k], rtol=0, atol=1e-10)
        _, ll = ctx.problem_eval(prob, r["live_u"][run])
        np.testing.assert_allclose(ll, r["live_logl"][run], rtol=0, atol=1e-10)
        assert (np.diff(r["dead_logl"][run, :k]) >= 0).all()
        # the reference's per-point bookkeeping (sampler.py:1165-1182): 'id' = live slot, 'it' = iteration
        # (from 1) at which the point was proposed, 'nc' = calls spent on its replacement.  Exact chain
        # property: a dead point of slot s was born when the previous occupant of s died (0: initial point)
        ids, its, ncs = (r[f][run, :k] for f in ("dead_id", "dead_it", "dead_nc"))
        assert ids.min() >= 0 and ids.max() < 300 and ncs.min() >= 1
        last = np.zeros(300, dtype=np.int64)
        for i in range(k):
            assert its[i] == last[ids[i]], (run, i)
            last[ids[i]] = i + 1
        np.testing.assert_array_equal(r["live_it"][run], last)
        # every call of the run is charged to exactly one death, except those after the last one and the
        # nlive calls of the initial points
        assert 0 <= int(r["ncall"][run]) - 300 - int(ncs.sum()) < 64 * 25 + 1
    assert m.ncall.shape == m.logl.shape and int(m.ncall.sum()) == int(sum(
        r["dead_nc"][q, :int(r["niter"][q])].sum() for q in range(8))) + 8 * 300
    sel = m.samples_run == 3
    np.testing.assert_array_equal(np.sort(m.samples_it[sel]), np.sort(np.concatenate(
        [r["dead_it"][3, :int(r["niter"][3])], r["live_it"][3]])))
    assert m.niter == int(r["niter"].sum()) + 8 * 300
    assert (np.diff(m.logl) >= 0).all()
    assert abs(m.logz[-1] - prob.logz_truth) < 5 * m.logzerr[-1] + 0.05
    w = m.importance_weights()
    mean = (w[:, None] * m.samples).sum(0)
    var = (w[:, None] * (m.samples - mean) ** 2).sum(0)
    # G5: unit-variance Gaussian posterior centred on 0 (prior is wide)
    assert np.abs(mean).max() < 0.15
    assert np.abs(var - 1.0).max() < 0.25
    # the merged evidence is at least as tight as a single run's
    assert m.logzerr[-1] < r["logzerr"].mean()


def test_sharded_merge_equals_single_process(ctx):
    """run_ensemble_merged_sharded (row blocks through the ragged gather) gives
    the same merged run as the single-process combiner."""
    from dynesty_amd import ensemble
    prob = inputs.problem("C1")
    kw = dict(nlive=200, queue_size=32, walks=23, bound="single", dlogz=0.3, max_iter=20000)
    a = ensemble.run_ensemble_merged(prob, 5, entropy=[12], **kw)
    b = ensemble.run_ensemble_merged_sharded(prob, 5, base_seed=12, **kw)
    assert a.niter == b.niter
    for k in ("ncall", "samples_id", "samples_it"):  # the per-point bookkeeping travels through the gather
        np.testing.assert_array_equal(a[k], b[k])
    np.testing.assert_array_equal(a.logl, b.logl)
    np.testing.assert_array_equal(a.samples_u, b.samples_u)
    np.testing.assert_array_equal(a.logz, b.logz)


def test_philox_proposals_in_the_device_loop(ctx):
    """sample='rwalk', rng='philox': the proposals of the device-resident loop drawn from hiprand Philox
    streams (throughput RNG mode).  Same statistics as the parity streams, deterministic, and independent of
    how the ensemble is sharded (the Philox subsequence is the global walker slot)."""
    prob = inputs.problem("G5")
    kw = dict(nlive=300, queue_size=64, walks=25, bound="multi", entropy=[17, 4], dlogz=0.05, rng="philox")
    r = ctx.ns_ensemble(prob, 16, **kw)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    se = lz.std(ddof=1) / 4.0
    assert abs(lz.mean() - prob.logz_truth) < 5 * se + 0.08, (lz.mean(), se)
    again = ctx.ns_ensemble(prob, 16, **kw)
    np.testing.assert_array_equal(again["logz"], lz)
    lo = ctx.ns_ensemble(prob, 8, first_run=0, **kw)
    hi = ctx.ns_ensemble(prob, 8, first_run=8, **kw)
    np.testing.assert_array_equal(np.concatenate([lo["logz"], hi["logz"]]), lz)
    # not the PCG64 streams
    pcg = ctx.ns_ensemble(prob, 16, **dict(kw, rng="pcg64"))
    assert (pcg["logz"] != lz).all()
    # (rslice / slice with rng='philox': tests/test_gpu_philox.py::test_philox_resident_loop_all_samplers)


def test_rebuild_beside_the_walk_gives_the_same_runs(monkeypatch):
    """DH_NS_OVERLAP=1 (measured, off by default): a run whose bound is due sits one fill out while its rebuild runs on
    a second stream beside the other runs' walkers.  Its own sequence is unchanged, so with PCG64 streams every run's
    result is bit-identical to the serial schedule's."""
    from dynesty_amd import _lib
    prob = inputs.problem("G5")
    # (the overlap exists in the late form of the forced update only: the default form switches it off)
    kw = dict(nlive=300, queue_size=64, walks=20, bound="multi", entropy=[5, 5], dlogz=0.1, forced_exact=False)
    a = _lib.Context(0).ns_ensemble(prob, 8, **kw)
    monkeypatch.setenv("DH_NS_OVERLAP", "1")
    b = _lib.Context(0).ns_ensemble(prob, 8, **kw)
    np.testing.assert_array_equal(a["logz"], b["logz"])
    np.testing.assert_array_equal(a["ncall"], b["ncall"])
    np.testing.assert_array_equal(a["nbound"], b["nbound"])
    assert b["nfills"] > a["nfills"]


@pytest.mark.parametrize("sample,kw", [("rwalk", dict(walks=20)), ("rslice", dict(slices=4)), ("unif", {})])
def test_waiting_for_the_rebuild_fill_changes_no_run(ctx, sample, kw):
    """rebuild_every = n: bounds are built every n-th fill and a run that is due waits.  Its own sequence is unchanged:
    with PCG64 streams every run's record is bit-identical for any n (1 = the reference schedule; 0 = the automatic
    choice), and so is the ensemble whatever its sharding."""
    prob = inputs.problem("G5")
    base = dict(nlive=300, queue_size=64, bound="multi", sample=sample, entropy=[8, 1], dlogz=0.1, **kw)
    ref = ctx.ns_ensemble(prob, 8, rebuild_every=1, **base)
    assert np.all(ref["status"] == 0)
    for n in (0, 2, 5, 16):
        r = ctx.ns_ensemble(prob, 8, rebuild_every=n, **base)
        for k in ("logz", "logzerr", "niter", "ncall", "nbound", "h"):
            np.testing.assert_array_equal(r[k], ref[k], err_msg=f"{k} at rebuild_every={n}")
    lo = ctx.ns_ensemble(prob, 4, first_run=0, rebuild_every=5, **base)
    hi = ctx.ns_ensemble(prob, 4, first_run=4, rebuild_every=5, **base)
    np.testing.assert_array_equal(np.concatenate([lo["logz"], hi["logz"]]), ref["logz"])


def test_forced_update_inside_the_fill_is_per_run_and_deterministic(ctx):
    """forced_exact=True (DH_NS_OPT_FORCED_EXACT): a run's result still depends on its own seed only (the forced
    rebuild of one run leaves the others' frames alone: an ensemble equals its shards) and it is deterministic.  When
    no start point ever lies outside the bound (one ellipsoid enlarged a thousandfold in volume) the runs keep the
    default form's number of bound updates -- none is forced -- while their regular bounds are built without the newest
    live point, as the reference builds them (sampler.py:771-772), so the trajectories are not the default form's."""
    from dynesty_amd import problems
    prob = problems.gauss_corr(13, 0.3, 5.0, "corr13")
    kw = dict(nlive=100, queue_size=16, walks=20, bound="multi", entropy=[3, 1, 4], dlogz=0.5, forced_exact=True)
    a = ctx.ns_ensemble(prob, 6, **kw)
    b = ctx.ns_ensemble(prob, 6, **kw)
    assert (a["status"] == 0).all()
    for key in ("logz", "niter", "ncall", "nbound"):
        np.testing.assert_array_equal(a[key], b[key])
    lo = ctx.ns_ensemble(prob, 2, first_run=0, **kw)
    hi = ctx.ns_ensemble(prob, 4, first_run=2, **kw)
    for key in ("logz", "niter", "ncall", "nbound"):
        np.testing.assert_array_equal(a[key], np.concatenate([lo[key], hi[key]]))
    late = ctx.ns_ensemble(prob, 6, **dict(kw, forced_exact=False))
    assert (late["logz"] != a["logz"]).any()      # (at 8 live points per dimension forced updates do occur)
    prob = inputs.problem("G5")
    kw1 = dict(nlive=200, queue_size=16, walks=20, bound="single", enlarge=1000.0, entropy=[9], dlogz=0.5)
    x = ctx.ns_ensemble(prob, 4, forced_exact=True, **kw1)
    y = ctx.ns_ensemble(prob, 4, forced_exact=False, **kw1)
    assert (x["status"] == 0).all() and (y["status"] == 0).all()
    assert abs(x["nbound"].mean() - y["nbound"].mean()) <= 1.0
    assert abs(x["logz"].mean() - y["logz"].mean()) < 0.5


def test_twenty_thousand_live_points(ctx):
    """the resident loop beyond nlive = 8192 (the queue consumption selects the K + 1
    smallest live points from the keys in global memory; the final live points are sorted in global memory): two runs
    of 20 000 live points on the 5-D correlated Normal, queue of 1024.  ln Z against the analytic value with the
    run's own error estimate (sqrt(H / N) ~ 0.02), and the dead points in order."""
    prob = inputs.problem("G5")
    r = ctx.ns_ensemble(prob, 2, 20000, 1024, walks=25, bound="multi", entropy=[20, 0, 0, 0], dlogz=0.01,
                        max_iter=600000, want_dead_logl=True)
    assert (r["status"] == 0).all()
    for i in range(2):
        n = int(r["niter"][i])
        assert n > 150000
        assert (np.diff(r["dead_logl"][i, :n]) >= 0).all()
        assert abs(r["logz"][i] - prob.logz_truth) < 5 * r["logzerr"][i] + 0.02, (r["logz"][i], prob.logz_truth)
        assert 0.01 < r["logzerr"][i] < 0.04


def test_boundary_flags_that_flag_nothing_change_nothing(ctx):
    """periodic=[] / reflective=[] (all coordinates hard) is the run without flags, bit for bit; a value that is not a
    DH_BC_* flag is refused; the flags are per call (the next call without them runs without them)."""
    from dynesty_amd import _lib
    prob = inputs.problem("G5")
    kw = dict(nlive=200, queue_size=32, walks=20, bound="multi", entropy=[4, 4], dlogz=0.5)
    a = ctx.ns_ensemble(prob, 4, **kw)
    b = ctx.ns_ensemble(prob, 4, periodic=[], reflective=[], **kw)
    c = ctx.ns_ensemble(prob, 4, periodic=[0, 2], **kw)
    d = ctx.ns_ensemble(prob, 4, **kw)
    for key in ("logz", "niter", "ncall"):
        np.testing.assert_array_equal(a[key], b[key])
        np.testing.assert_array_equal(a[key], d[key])
    assert (c["status"] == 0).all()
    bad = np.full(5, 7, dtype=np.int8)  # not a DH_BC_* flag
    assert ctx.lib.dh_ns_set_boundary(ctx.handle, 5, bad.ctypes.data) < 0
