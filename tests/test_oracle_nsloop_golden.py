"""oracle/nested_ref.py (queue consumption, evidence recurrence, final live points) against what a
REAL NestedSampler run recorded (tests/golden/nsloop.npz, tools/make_golden.py nsloop)."""
import os

import numpy as np
import pytest

from oracle import nested_ref as R

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nsloop.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(G)


def state_at(g, tag):
    s = R.RunState(int(g["nlive"]))
    s.logz, s.logzvar, s.h = (float(g[f"fill_{tag}/state_{k}"]) for k in ("logz", "logzvar", "h"))
    s.logvol = float(g[f"fill_{tag}/state_logvol"])
    s.loglstar = float(g[f"fill_{tag}/state_logl"])
    s.it = int(g[f"fill_{tag}/it0"])
    s.plateau_mode = bool(g[f"fill_{tag}/state_plateau_mode"])
    s.plateau_counter = int(g[f"fill_{tag}/state_plateau_counter"])
    s.plateau_logdvol = float(g[f"fill_{tag}/state_plateau_logdvol"])
    return s


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_consume_queue_matches_the_reference_run(g, tag):
    s = state_at(g, tag)
    live = g[f"fill_{tag}/live_logl"].copy()
    out = R.consume_queue(live, g[f"fill_{tag}/q_logl"], g[f"fill_{tag}/q_ncalls"], s, float(g["dlogz"]))
    want = g[f"fill_{tag}/dead_logl"]
    np.testing.assert_array_equal(out["dead_logl"], want)
    np.testing.assert_array_equal(out["dead_slot"], g[f"fill_{tag}/dead_slot"])
    assert out["stopped"] == bool(g[f"fill_{tag}/is_last"])
    if len(want):
        # bit-identical: the same scipy / numpy calls in the same order
        assert s.logz == g[f"fill_{tag}/logz"][-1]
        assert s.logzvar == g[f"fill_{tag}/logzvar"][-1]
        assert s.h == g[f"fill_{tag}/h"][-1]
    assert s.it == int(g[f"fill_{tag}/it0"]) + len(want)
    # some queue entries of a mid-run fill are stale (fewer deaths than entries)
    assert len(want) <= int(g["K"])


def test_whole_run_recurrence(g):
    """All fills chained: the recurrence over the whole run (what the reference accumulates while
    sampling, plateau steps included) ends bit-identical to the real run."""
    nlive, nf = int(g["nlive"]), int(g["nfills"])
    s = R.RunState(nlive)
    live = g["fills/live_logl0"].copy()
    live_it = np.zeros(nlive, dtype=np.int64)
    dead, ids, its, ncs = [], [], [], []
    carry = 0
    for f in range(nf):
        out = R.consume_queue(live, g["fills/q_logl"][f], g["fills/q_ncalls"][f], s, float(g["dlogz"]),
                              live_it=live_it)
        dead.append(out["dead_logl"])
        ids.append(out["dead_slot"])
        its.append(out["dead_it"])
        nc = out["dead_nc"].copy()
        if len(nc):  # entries popped after the previous fill's last death belong to this fill's first
            nc[0] += carry
            carry = 0
        carry += out["nc_carry"]
        ncs.append(nc)
        assert out["stopped"] == (f == nf - 1)
    dead = np.concatenate(dead)
    niter = int(g["niter"])
    np.testing.assert_array_equal(dead, g["run/logl"][:niter])
    # the reference's per-point id / it / nc (saved_run; what merge_runs carries as samples_id /
    # samples_it / ncall): dead points, then the final live points (id = slot, nc = 1)
    np.testing.assert_array_equal(np.concatenate(ids), g["run/id"][:niter])
    np.testing.assert_array_equal(np.concatenate(its), g["run/it"][:niter])
    np.testing.assert_array_equal(np.concatenate(ncs), g["run/nc"][:niter])
    order = np.argsort(live)
    np.testing.assert_array_equal(order, g["run/id"][niter:])
    np.testing.assert_array_equal(live_it[order], g["run/it"][niter:])
    assert (g["run/nc"][niter:] == 1).all()
    assert s.ncall + int(g["run/ncall_init"]) == int(g["run/ncall"])
    logz, logzvar, h = R.add_live_points(live, s)
    assert logz == float(g["rec/logz_final"])
    assert logzvar == float(g["rec/logzvar_final"])
    assert h == float(g["rec/h_final"])
    np.testing.assert_array_equal(np.sort(live), g["run/logl"][niter:])


def test_compute_integrals_is_what_results_reports(g):
    niter, nlive = int(g["niter"]), int(g["nlive"])
    logl = g["run/logl"]
    # up to the first likelihood plateau (rwalk duplicates) the volumes are the plain ladder
    first = int(g["run/first_plateau_it"])
    np.testing.assert_array_equal(R.static_run_logvol(niter, nlive)[:first], g["run/logvol"][:first])
    _, logz, logzvar, h = R.compute_integrals(logl, g["run/logvol"])
    np.testing.assert_array_equal(logz, g["run/logz"])
    np.testing.assert_array_equal(logzvar, g["run/logzvar"])
    np.testing.assert_array_equal(h, g["run/h"])
    assert logz[-1] == float(g["run/logz_final"]) and h[-1] == float(g["run/h_final"])
    np.testing.assert_allclose(np.sqrt(logzvar[-1]), float(g["run/logzerr_final"]), rtol=1e-15)
    lz = float(logz[-1]); hh = float(h[-1])
    # without the plateau steps (the device loop's simplification) the result moves by < 1e-4
    lz2, lzerr2, hh2 = R.final_results(logl[:niter], logl[niter:], nlive)
    assert abs(lz2 - lz) < 1e-4 and abs(lzerr2 - float(g["run/logzerr_final"])) < 1e-4
    # the recurrence and the final recomputation agree on ln Z and H to rounding; var[ln Z]
    # differs in the final-live-point part (different normalisation of the partial H's)
    np.testing.assert_allclose(float(g["rec/logz_final"]), lz, rtol=0, atol=1e-11)
    np.testing.assert_allclose(float(g["rec/h_final"]), hh, rtol=1e-10)


def test_integrate_full_is_compute_integrals(g):
    """dynesty_amd.nested._integrate_full (used by the host driver and the ensemble combiner)
    against the reference's recorded per-point history."""
    from dynesty_amd import nested
    _, logz, h, logzvar = nested._integrate_full(g["run/logl"], g["run/logvol"])
    np.testing.assert_allclose(logz, g["run/logz"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(h, g["run/h"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(logzvar, g["run/logzvar"], rtol=1e-8, atol=1e-14)


def test_static_run_logvol_reproduces_the_plateau_steps_of_the_real_run(g):
    """dynesty_amd.nested.static_run_logvol (the host driver's ln X of every point: the constant ladder, the
    reference's plateau steps where live points share a log-likelihood, sampler.py:1112-1127 / 813-830) from the
    run's own per-point record (logl, it) against the real run's recorded volumes -- dead and final live points."""
    from dynesty_amd import nested
    niter, nlive = int(g["niter"]), int(g["nlive"])
    logl, it = g["run/logl"], g["run/it"]
    dead_lv, live_lv = nested.static_run_logvol(logl[:niter], it[:niter], logl[niter:], it[niter:], nlive)
    np.testing.assert_allclose(dead_lv, g["run/logvol"][:niter], rtol=1e-13, atol=0)
    np.testing.assert_allclose(live_lv, g["run/logvol"][niter:], rtol=1e-13, atol=0)
    first = int(g["run/first_plateau_it"])
    dlv = np.log((nlive + 1.0) / nlive)
    assert np.abs(dead_lv[:first] + dlv * np.arange(1, first + 1)).max() < 1e-12
    assert abs(dead_lv[-1] + niter * dlv) > 1e-6  # the plateau steps are really there
    # and with them the product's integration gives the run's final record
    _, logz, h, logzvar = nested._integrate_full(logl, np.concatenate([dead_lv, live_lv]))
    np.testing.assert_allclose(logz[-1], float(g["run/logz_final"]), rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.sqrt(logzvar[-1]), float(g["run/logzerr_final"]), rtol=1e-9)


def test_logvol_from_record_is_the_real_runs(g):
    """The oracle's replay of the volume bookkeeping from a finished run's record (used to check whole device
    runs that contain plateaus) against the real run's recorded volumes."""
    niter, nlive = int(g["niter"]), int(g["nlive"])
    final = np.empty(nlive)
    final[g["run/id"][niter:]] = g["run/logl"][niter:]
    lv = R.logvol_from_record(g["run/logl"][:niter], g["run/id"][:niter], final, nlive)
    np.testing.assert_array_equal(lv, g["run/logvol"])
