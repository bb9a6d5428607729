"""The device-resident run loop's bookkeeping (csrc/ns.hip: ns_consume / ns_finish) pinned to the
oracle's restatement of the reference loop (oracle/nested_ref.py, itself bit-identical to a recorded
real NestedSampler run: tests/test_oracle_nsloop_golden.py):

  * dh_ns_consume replays queue fills the REAL reference run recorded (tests/golden/nsloop.npz):
    dead points, slots and replacements exact, ln Z / H / var ln Z to 1e-11
  * chains of synthetic fills over several runs vs oracle.consume_queue (incl. the dlogz stop)
  * a whole device run: the reported ln Z, error and information equal compute_integrals
    (what the reference's Results holds) over the run's own dead + live log-likelihoods to 1e-9
"""
import os

import numpy as np
import pytest

import inputs
from oracle import nested_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nsloop.npz")


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def pack(states):
    return np.array([[s.logvol, s.logz, s.h, s.logzvar, s.loglstar, s.it, s.ncall, 0.0] for s in states])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_replay_of_a_real_reference_fill(ctx, tag):
    g = np.load(G)
    assert int(g[f"fill_{tag}/it0"]) + len(g[f"fill_{tag}/dead_logl"]) <= int(g["run/first_plateau_it"])
    nlive = int(g["nlive"])
    s = R.RunState(nlive)
    s.logz, s.logzvar, s.h = (float(g[f"fill_{tag}/state_{k}"]) for k in ("logz", "logzvar", "h"))
    s.logvol, s.loglstar = float(g[f"fill_{tag}/state_logvol"]), float(g[f"fill_{tag}/state_logl"])
    s.it = int(g[f"fill_{tag}/it0"])
    live = g[f"fill_{tag}/live_logl"][None, :].copy()
    state = pack([s])
    out = ctx.ns_consume(live, g[f"fill_{tag}/q_logl"][None], g[f"fill_{tag}/q_ncalls"][None], state,
                         float(g["dlogz"]))
    np.testing.assert_array_equal(out["dead_logl"][0], g[f"fill_{tag}/dead_logl"])
    np.testing.assert_array_equal(out["dead_slot"][0], g[f"fill_{tag}/dead_slot"])
    assert not out["stopped"][0]
    np.testing.assert_allclose(state[0, 1], g[f"fill_{tag}/logz"][-1], rtol=0, atol=1e-11)
    np.testing.assert_allclose(state[0, 2], g[f"fill_{tag}/h"][-1], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(state[0, 3], g[f"fill_{tag}/logzvar"][-1], rtol=1e-9, atol=1e-15)
    assert int(state[0, 5]) == s.it + len(g[f"fill_{tag}/dead_logl"])
    assert int(state[0, 6]) == int(g[f"fill_{tag}/q_ncalls"].sum())
    # the live set after the fill: the oracle's
    ref_live = g[f"fill_{tag}/live_logl"].copy()
    R.consume_queue(ref_live, g[f"fill_{tag}/q_logl"], g[f"fill_{tag}/q_ncalls"], s, float(g["dlogz"]))
    np.testing.assert_array_equal(live[0], ref_live)


@pytest.mark.parametrize("nlive,K,runs", [(300, 100, 5), (2000, 512, 3), (64, 17, 4), (100, 256, 3), (1000, 2048, 2),
                                          (5000, 1024, 2), (8192, 2048, 2), (9000, 700, 2), (20000, 2048, 2),
                                          (40000, 1500, 1)])
def test_fill_chains_vs_oracle(ctx, nlive, K, runs):
    """Several runs, consecutive fills, each run's state carried from fill to fill on both sides.
    The queue mixes entries above and below the moving threshold (stale ones are discarded);
    dlogz is set so that the stopping rule fires inside one of the last fills.  The last four shapes (round 4) are
    beyond what the kernel holds in LDS at once -- nlive = 8192 together with K = 2048, nlive > 8192 -- and go
    through its selection of the K + 1 smallest live points (ns_consume_compact)."""
    rng = np.random.default_rng(nlive + K)
    # a Gaussian-like likelihood in 6-D: logl = -r^2/2, live points uniform in a ball of radius 6
    def draw(n, rad):
        return -0.5 * (rad * rng.random(n) ** (1 / 6.0)) ** 2
    live = np.stack([draw(nlive, 6.0) for _ in range(runs)])
    ref_live = live.copy()
    states = [R.RunState(nlive) for _ in range(runs)]
    state = pack(states)
    # the reference's per-point bookkeeping ('it', 'nc' of saved_run) rides along
    live_it = np.zeros((runs, nlive), dtype=np.int32)
    ref_it = np.zeros((runs, nlive), dtype=np.int64)
    dlogz = 0.5
    done = np.zeros(runs, bool)
    plat = np.zeros((runs, 2))
    nfill = 0
    while not done.all() and nfill < 400:
        nfill += 1
        ql = np.empty((runs, K))
        for r in range(runs):
            thr = ref_live[r].min()
            rad = np.sqrt(-2 * thr)
            # proposals drawn inside a ball slightly LARGER than the contour: ~25% are stale at birth,
            # more become stale while the fill is consumed
            ql[r] = draw(K, rad * 1.05)
        qn = rng.integers(1, 60, size=(runs, K)).astype(np.int32)
        act = ~done
        if not act.all():  # finished runs leave the ensemble (as MODE_DONE runs do on the device)
            live_a, state_a = np.ascontiguousarray(live[act]), np.ascontiguousarray(state[act])
            it_a, plat_a = np.ascontiguousarray(live_it[act]), np.ascontiguousarray(plat[act])
        else:
            live_a, state_a, it_a, plat_a = live, state, live_it, plat
        out = ctx.ns_consume(live_a, ql[act], qn[act], state_a, dlogz, live_it=it_a, plateau=plat_a)
        live[act], state[act], live_it[act], plat[act] = live_a, state_a, it_a, plat_a
        for i, r in enumerate(np.flatnonzero(act)):
            ref = R.consume_queue(ref_live[r], ql[r], qn[r], states[r], dlogz, plateau=True, live_it=ref_it[r])
            np.testing.assert_array_equal(out["dead_it"][i], ref["dead_it"])
            np.testing.assert_array_equal(out["dead_nc"][i], ref["dead_nc"])
            np.testing.assert_array_equal(live_it[r], ref_it[r])
            np.testing.assert_array_equal(out["dead_logl"][i], ref["dead_logl"])
            np.testing.assert_array_equal(out["dead_slot"][i], ref["dead_slot"])
            np.testing.assert_array_equal(out["dead_src"][i], ref["dead_src"])
            assert bool(out["stopped"][i]) == ref["stopped"]
            s = states[r]
            # -(it * dlv) on the device against `it` subtractions in the oracle (up to 3e5 of them at nlive = 40 000)
            np.testing.assert_allclose(state[r, 0], s.logvol, rtol=1e-12 if nlive <= 5000 else 2e-11)
            np.testing.assert_allclose(state[r, 1], s.logz, rtol=0, atol=1e-10)
            np.testing.assert_allclose(state[r, 2], s.h, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(state[r, 3], s.logzvar, rtol=1e-8, atol=1e-14)
            assert int(state[r, 5]) == s.it
            if not ref["stopped"]:
                assert int(state[r, 6]) == s.ncall
            else:
                # the reference stops BEFORE popping the next entry; the device charges the calls up to
                # the replacement of the last death: identical
                assert int(state[r, 6]) == s.ncall
            np.testing.assert_array_equal(live[r], ref_live[r])
            assert state[r, 7] == ref_live[r].min()
            done[r] = ref["stopped"]
    assert done.all() and nfill > 3


def test_whole_run_results_are_compute_integrals(ctx):
    """dh_ns_ensemble's record (ln Z, sqrt var, H) = the reference's final recomputation
    (compute_integrals, sampler.py:1342-1348) over the run's own dead and live log-likelihoods.  (The volumes are
    replayed from the record with the reference's bookkeeping: with 25 steps per proposal an rwalk walker still hands
    back its start point now and then -- an exact tie among the live points, i.e. a plateau step, sampler.py:1112-1127;
    round 5's default protocol met one in run 2 of this very ensemble, where the plain ladder is 1e-5 off in ln Z.)"""
    prob = inputs.problem("G5")
    r = ctx.ns_ensemble(prob, 6, 300, 64, walks=25, bound="multi", entropy=[31], dlogz=0.05,
                        max_iter=40000, want_samples=True)
    assert np.all(r["status"] == 0)
    nties = 0
    for i in range(6):
        n = int(r["niter"][i])
        dead, ids = r["dead_logl"][i, :n], r["dead_id"][i, :n]
        ties = len(np.unique(np.concatenate([dead, r["live_logl"][i]]))) < n + 300
        nties += ties
        lv = R.logvol_from_record(dead, ids, r["live_logl"][i], 300)
        if not ties:  # no plateau met: the plain ladder
            np.testing.assert_allclose(lv, R.static_run_logvol(n, 300), rtol=0, atol=1e-9)
        logl = np.concatenate([dead, np.sort(r["live_logl"][i])])
        _, lzs, lzvars, hs = R.compute_integrals(logl, lv)
        lz, lzerr, h = float(lzs[-1]), float(np.sqrt(lzvars[-1])), float(hs[-1])
        np.testing.assert_allclose(r["logz"][i], lz, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["logzerr"][i], lzerr, rtol=1e-7)
        np.testing.assert_allclose(r["h"][i], h, rtol=1e-9)
        if ties:
            continue
        # and the stopping rule held exactly at the last dead point, not before (recurrence values)
        s = R.RunState(300)
        for lnew in dead:
            s.logvol -= s.dlv
            _, s.logz, s.logzvar, s.h = R.progress_integration(s.loglstar, lnew, s.logz, s.logzvar,
                                                               s.logvol, s.dlv, s.h)
            s.loglstar = lnew
        lmax = r["live_logl"][i].max()
        assert np.logaddexp(0, lmax + s.logvol - s.logz) < 0.05
    assert nties < 6


def test_whole_run_with_plateaus_is_compute_integrals(ctx):
    """rwalk with two steps per proposal hands back its start point about a quarter of the time: duplicates among
    the live points, i.e. likelihood plateaus all along the run.  The device-resident loop's record must then equal
    the reference's final recomputation over the run's points WITH the plateau volume steps (oracle replay of the
    bookkeeping from the run's own record: dead points + slots, final live points)."""
    prob = inputs.problem("G5")
    r = ctx.ns_ensemble(prob, 4, 300, 64, walks=2, bound="multi", entropy=[41], dlogz=0.05, max_iter=40000,
                        want_samples=True)
    assert np.all(r["status"] == 0)
    for i in range(4):
        n = int(r["niter"][i])
        dead, ids = r["dead_logl"][i, :n], r["dead_id"][i, :n]
        assert len(np.unique(np.concatenate([dead, r["live_logl"][i]]))) < n  # duplicates: the run met plateaus
        lv = R.logvol_from_record(dead, ids, r["live_logl"][i], 300)
        assert np.abs(lv[:n] - R.static_run_logvol(n, 300)[:n]).max() > 1e-5  # and its volumes left the ladder
        logl = np.concatenate([dead, np.sort(r["live_logl"][i])])
        _, logz, logzvar, h = R.compute_integrals(logl, lv)
        np.testing.assert_allclose(r["logz"][i], logz[-1], rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["logzerr"][i], np.sqrt(logzvar[-1]), rtol=1e-7)
        np.testing.assert_allclose(r["h"][i], h[-1], rtol=1e-9)
        # the plain ladder would NOT have given this ln Z
        lz2, _, _ = R.final_results(dead, r["live_logl"][i], 300)
        assert abs(lz2 - r["logz"][i]) > 1e-8


@pytest.mark.parametrize("nlive", [300, 9000])
@pytest.mark.parametrize("nlev", [40, 400, 3000])
def test_ties_die_lowest_slot_first_and_take_the_plateau_steps(ctx, nlev, nlive):
    """Equal log-likelihoods among the live points (rwalk hands back its start point when no step was accepted) die
    lowest slot first -- np.argmin's rule in the reference (sampler.py:1107) -- also when the tie is between an original
    live point and a replacement made earlier in the same fill, and their deaths take the reference's PLATEAU volume
    steps (sampler.py:1112-1127, 1190-1193: a constant volume step X / (N + 1) per death while the plateau lasts)
    instead of ln((N + 1) / N).  Values drawn from a small set force many plateaus, also across fill boundaries; the
    oracle runs with plateau=True (the restatement that is bit-identical to the recorded real run).  40 levels: groups
    of dozens of equal proposals (the serial walk takes those fills); 400 / 3000 levels: pairs and triples, settled
    inside the parallel walk.  nlive = 9000 (round 4): the same through the selection of the K + 1 smallest live
    points, where with 40 levels the whole subset and hundreds of points outside it share one value (ties at the
    threshold are taken lowest slot first, and the plateau's multiplicity counts the points left outside)."""
    rng = np.random.default_rng(3 + nlev)
    K, runs = 200, 4 if nlive == 300 else 2
    vals = -np.sort(rng.random(nlev) * 20)  # nlev distinct levels
    live = vals[rng.integers(0, nlev, size=(runs, nlive))].copy()
    ref_live = live.copy()
    states = [R.RunState(nlive) for _ in range(runs)]
    state = pack(states)
    plat = np.zeros((runs, 2))
    live_it = np.zeros((runs, nlive), dtype=np.int32)
    ref_it = np.zeros((runs, nlive), dtype=np.int64)
    nplateau_fills = 0
    for fill in range(6):
        ql = vals[rng.integers(0, nlev, size=(runs, K))] + (fill * 0.0)
        ql[:, ::3] += 0.5  # some entries off the grid
        qn = rng.integers(1, 9, size=(runs, K)).astype(np.int32)
        out = ctx.ns_consume(live, ql, qn, state, None if False else 1e-300, live_it=live_it, plateau=plat)
        for r in range(runs):
            ref = R.consume_queue(ref_live[r], ql[r], qn[r], states[r], 1e-300, plateau=True, live_it=ref_it[r])
            np.testing.assert_array_equal(out["dead_logl"][r], ref["dead_logl"])
            np.testing.assert_array_equal(out["dead_slot"][r], ref["dead_slot"])
            np.testing.assert_array_equal(out["dead_src"][r], ref["dead_src"])
            np.testing.assert_array_equal(out["dead_it"][r], ref["dead_it"])
            np.testing.assert_array_equal(out["dead_nc"][r], ref["dead_nc"])
            np.testing.assert_array_equal(live[r], ref_live[r])
            np.testing.assert_array_equal(live_it[r], ref_it[r])
            assert state[r, 7] == ref_live[r].min()
            sr = states[r]
            # the volume really left the ln((N + 1) / N) ladder
            assert abs(sr.logvol + sr.it * sr.dlv) > (1e-6 if nlev <= 400 else 0.0)
            np.testing.assert_allclose(state[r, 0], sr.logvol, rtol=1e-12)
            np.testing.assert_allclose(state[r, 1], sr.logz, rtol=0, atol=1e-10)
            np.testing.assert_allclose(state[r, 2], sr.h, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(state[r, 3], sr.logzvar, rtol=1e-8, atol=1e-14)
            assert int(plat[r, 0]) == (sr.plateau_counter if sr.plateau_mode else 0)
            if sr.plateau_mode:
                nplateau_fills += 1
                np.testing.assert_allclose(plat[r, 1], sr.plateau_logdvol, rtol=1e-12)
    assert nplateau_fills > 0 or nlev > 400  # plateaus were carried across fill boundaries


def test_replay_of_the_whole_real_run_with_its_plateaus(ctx):
    """Every queue fill the REAL reference run recorded (tests/golden/nsloop.npz), chained through dh_ns_consume with the
    plateau mode carried from fill to fill: the same dead points in the same slots, and after every fill the run's own
    recorded ln X (plateau steps included: the run meets its first plateau at iteration `first_plateau_it`) and ln Z
    (the information and var[ln Z] of the recurrence are held to the oracle in the test above; the run's own record
    of them is the final recomputation, normalised differently)."""
    g = np.load(G)
    nlive, nf, niter = int(g["nlive"]), int(g["nfills"]), int(g["niter"])
    assert int(g["run/first_plateau_it"]) < niter  # the recorded run does contain plateaus
    live = g["fills/live_logl0"][None, :].copy()
    state = pack([R.RunState(nlive)])
    plat = np.zeros((1, 2))
    live_it = np.zeros((1, nlive), dtype=np.int32)
    dead, ids, its = [], [], []
    for f in range(nf):
        out = ctx.ns_consume(live, g["fills/q_logl"][f][None], g["fills/q_ncalls"][f][None], state, float(g["dlogz"]),
                             live_it=live_it, plateau=plat)
        dead.append(out["dead_logl"][0])
        ids.append(out["dead_slot"][0])
        its.append(out["dead_it"][0])
        assert bool(out["stopped"][0]) == (f == nf - 1)
        it = int(state[0, 5])
        if it:
            np.testing.assert_allclose(state[0, 0], g["run/logvol"][it - 1], rtol=1e-12)
            np.testing.assert_allclose(state[0, 1], g["run/logz"][it - 1], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(np.concatenate(dead), g["run/logl"][:niter])
    np.testing.assert_array_equal(np.concatenate(ids), g["run/id"][:niter])
    np.testing.assert_array_equal(np.concatenate(its), g["run/it"][:niter])
    # the volumes did leave the constant ladder
    dlv = np.log((nlive + 1.0) / nlive)
    assert abs(state[0, 0] + niter * dlv) > 1e-7


def test_presorted_slot_order_equals_the_sort_in_place():
    """Round 6: in an rwalk fill the live points' slot order comes from presort workgroups in front of the generator pass
    (PCG64 items and Philox rows alike) and ns_consume reads it; DH_NS_PRESORT=0 keeps the sort inside ns_consume.  The
    same order either way: whole runs iteration for iteration, both RNG modes, nlive at and off the sort's power of two."""
    import os
    from dynesty_amd import _lib, problems
    c = _lib.Context(0)
    prob = problems.gauss_corr(10, 0.4, 5.0, "corr10")
    for nlive, K, rng in ((512, 128, "pcg64"), (700, 256, "pcg64"), (2000, 512, "philox")):
        outs = []
        for v in ("1", "0"):
            os.environ["DH_NS_PRESORT"] = v
            try:
                outs.append(c.ns_ensemble(prob, 5, nlive, K, walks=20, bound="multi", entropy=[9], want_dead_logl=True, rng=rng))
            finally:
                del os.environ["DH_NS_PRESORT"]
        a, b = outs
        assert (a["status"] == 0).all() and int(a["niter"].min()) > nlive
        for k in ("logz", "logzerr", "niter", "ncall", "nbound"):
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)
        for r, n in enumerate(a["niter"]):
            np.testing.assert_array_equal(a["dead_logl"][r][:n], b["dead_logl"][r][:n])
