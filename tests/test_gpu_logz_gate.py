"""The north_star's evidence gate -- "logZ within +-0.05 of reference" -- against ensembles of the REAL
reference (dynesty 3.0.0 run in the build container; tests/golden/c2_logz_ref.json by tools/ref_c2_runs.py,
tests/golden/c4_logz_ref.json by tools/ref_rslice_bias.py / ref_c4_runs.py) at the queue sizes the device
paths use.  An ensemble mean is only known to its standard error, so the bound is
max(0.05, 3 * sqrt(se_device^2 + se_reference^2)).

Reference facts these files hold (and that the gate therefore carries):
  C3 (eggbox 2-D, nlive 5000, multi/rslice):  see tests/golden/c3_logz_ref.json (16 runs each at K = 1 and
     SerialPool(1024); about 235.89 +- 0.01, test-suite truth 235.856);
  C2 (25-D rho=0.4 Normal, nlive 2000, multi/rwalk):  K=1 -57.485 +- 0.026, K=512 -57.493 +- 0.023,
     K=2000 -57.266 +- 0.028  (analytic -57.5646): the queue-size bias at K = nlive is the reference's own;
  C4 (200-D, nlive 4000, single/rslice):  K=1 -250.820 +- 0.021 (9 runs), SerialPool(1000) -250.960 +- 0.028
     (9 runs), SerialPool(4000) -251.30 (2 runs); analytic -253.10: the +2.2 offset, and the drift with the queue
     size, are the reference's own.
"""
import json
import math
import os

import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


def bound(se_a, se_b):
    return max(0.05, 3.0 * math.sqrt(se_a * se_a + se_b * se_b))


@pytest.mark.parametrize("K,ref_key,rng", [(512, "K512", "pcg64"), (512, "K1", "pcg64"), (2000, "K2000", "pcg64"),
                                           (512, "K512", "philox")])
def test_c2_device_ensemble_vs_reference_ensemble(ctx, K, ref_key, rng):
    ref = json.load(open(os.path.join(GOLD, "c2_logz_ref.json")))["ensembles"][ref_key]
    prob = inputs.problem("C2")
    r = ctx.ns_ensemble(prob, 64, 2000, K, walks=45, bound="multi", entropy=[2026, K], dlogz=0.01, rng=rng)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    mean, se = lz.mean(), lz.std(ddof=1) / math.sqrt(len(lz))
    assert abs(mean - ref["mean"]) < bound(se, ref["se"]), (mean, se, ref["mean"], ref["se"])
    # the reference's scatter and error estimate are the device's
    assert 0.6 < lz.std(ddof=1) / ref["std"] < 1.6
    assert abs(r["logzerr"].mean() - ref["mean_logzerr"]) < 0.01
    # work per run within 5 % of the reference's (same proposals per iteration, same stopping rule)
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.02
    if ref["K"] == K:  # (a queue of K costs the discarded stale proposals: only comparable at equal K)
        assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.08


def test_c2_throughput_k_is_not_the_gate_k(ctx):
    """K = 2000 (one whole bound-update interval in flight, the tap-A launch shape) is biased by +0.2 --
    in the reference exactly as on the device -- so the end-to-end legs run at K = 512."""
    g = json.load(open(os.path.join(GOLD, "c2_logz_ref.json")))["ensembles"]
    assert g["K2000"]["mean"] - g["K512"]["mean"] > 0.15
    assert abs(g["K512"]["mean"] - g["K1"]["mean"]) < bound(g["K512"]["se"], g["K1"]["se"])


@pytest.mark.parametrize("K,ref_key", [(1024, "K1024"), (1024, "K1")])
def test_c3_device_ensemble_vs_reference_ensemble(ctx, K, ref_key):
    """BASELINE C3 at full size -- eggbox 2-D, nlive 5000, MultiEllipsoid (13-15 ellipsoids), rslice x 5 --
    through the device-resident loop: 32 runs against 16 runs of the real reference at the same queue size
    (SerialPool(1024)) and against its serial ensemble (tests/golden/c3_logz_ref.json, tools/ref_c3_runs.py).
    Driver followed: sampler.py:690-778, 1214-1356."""
    ref = json.load(open(os.path.join(GOLD, "c3_logz_ref.json")))["ensembles"][ref_key]
    assert ref["n"] >= 16
    prob = inputs.problem("C3")
    r = ctx.ns_ensemble(prob, 32, 5000, K, bound="multi", sample="rslice", slices=5, entropy=[2026, 3], dlogz=0.01)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    mean, se = lz.mean(), lz.std(ddof=1) / math.sqrt(len(lz))
    assert abs(mean - ref["mean"]) < bound(se, ref["se"]), (mean, se, ref["mean"], ref["se"])
    assert 0.5 < lz.std(ddof=1) / ref["std"] < 2.0
    assert abs(r["logzerr"].mean() - ref["mean_logzerr"]) < 0.005
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.02
    if ref["K"] == K:
        assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.08


def test_c4_device_run_vs_reference_runs(ctx):
    """BASELINE C4 to convergence through the wide-D path (host loop over device calls) against the two
    converged runs of the real reference."""
    from dynesty_amd import backend, nested, problems
    ref = json.load(open(os.path.join(GOLD, "c4_logz_ref.json")))
    k1 = [r for r in ref["runs"] if r["K"] == 1]
    ref_mean = float(np.mean([r["logz"] for r in k1]))
    ref_err = float(np.mean([r["logzerr"] for r in k1]))
    assert abs(ref_mean - ref["truth"]) > 2.0  # the reference's own offset at these settings
    prob = problems.gauss_normal_prior(200, "C4")
    backend.set_backend(ctx)
    try:
        r = nested.run_static(prob, nlive=4000, bound='single', sample='rslice', slices=203, queue_size=1000,
                              rstate=np.random.default_rng(21), dlogz=0.01)
    finally:
        backend.set_backend(None)
    # one run against the mean of two: sigma^2 = err^2 (1 + 1/2)
    assert abs(r.logz - ref_mean) < 3.0 * ref_err * math.sqrt(1.5), (r.logz, ref_mean)
    assert abs(r.logzerr - ref_err) < 0.01
    assert abs(r.niter / np.mean([x["niter"] for x in k1]) - 1) < 0.05
    # and against the reference at the SAME queue size (SerialPool(1000)): one run against one run
    for x in (q for q in ref["runs"] if q["K"] == 1000):
        assert abs(r.logz - x["logz"]) < 3.0 * math.hypot(r.logzerr, x["logzerr"]), (r.logz, x["logz"])
        assert abs(r.niter / x["niter"] - 1) < 0.06


def bound2(se_a, se_b):
    """north_star's +-0.05, or two combined standard errors where the ensembles cannot resolve that (VERDICT round 3
    item 5: 3 sigma was all the nine serial reference runs of round 3 allowed; there are 25+ now)."""
    return max(0.05, 2.0 * math.sqrt(se_a * se_a + se_b * se_b))


@pytest.mark.parametrize("K,ref_key,runs", [(128, "K1", 32), (128, "K128", 32), (1000, "K1000", 8)])
def test_c4_device_resident_loop_vs_reference_ensembles(ctx, K, ref_key, runs):
    """BASELINE C4 through the device-resident loop (dh_ns_ensemble at 200-D: wave-per-walker rslice kernels with
    per-run thresholds, masked multi-workgroup Ellipsoid.update) against the converged ensembles of the real reference
    (tests/golden/c4_logz_ref.json: 9 serial runs -250.820 +- 0.021, 9 runs with SerialPool(1000) -250.960 +- 0.028;
    45-70 minutes per run): at the reference's own queue size, and -- with the small queue the bench's C4 leg uses --
    against the SERIAL ensemble (ln Z drifts down with the queue size in the reference exactly as on the device:
    device 128 / 256 / 512 / 1000 -> -250.874 / -250.890 / -250.928 / -250.979, 16 runs each).
    Round 4: the serial reference ensemble has 32 runs (-250.820 +- 0.014; 57 min each), the device at the bench's
    K = 128 gives -250.864 +- 0.009 with 64 runs (K = 16 / 32 / 64: -250.854 / -250.847 / -250.855: flat below 128):
    0.044 apart, inside north_star's +-0.05 at the means but 2.7 combined sigma from zero -- the queue (any K >= 16)
    shifts ln Z by about -0.03 against the serial sampler, as it does in the reference (K = 1000: -0.14): at K = 4 the
    device gives -250.828 +- 0.008 (64 runs, 4.2 s per run; profiles/r04/c4_ksweep_64runs.jsonl), 0.008 from the serial
    reference.  The gate is max(0.05, 2 sigma) there.  And at the bench's own queue size the real reference
    (SerialPool(128), 80 min per run) gives -250.862 +- 0.015 (n = 20): the device's -250.864 +- 0.009 is 0.002 +- 0.017
    from it, and the reference's own K = 128 sits 0.042 +- 0.020 below its serial ensemble -- the queue's effect, in both."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "c4_logz_ref.json")))["ensembles"][ref_key]
    assert ref["n"] >= (20 if ref_key == "K1" else 4)  # (K128: 4 runs when first committed)
    prob = problems.gauss_normal_prior(200, "C4")
    r = ctx.ns_ensemble(prob, runs, 4000, K, bound='single', sample='rslice', slices=203, entropy=[21, K], dlogz=0.01,
                        max_iter=250000)
    assert (r["status"] == 0).all()
    lz = r["logz"]
    mean, se = lz.mean(), lz.std(ddof=1) / math.sqrt(runs)
    gate = bound2 if ref_key == "K1" else bound
    assert abs(mean - ref["mean"]) < gate(se, ref["se"]), (mean, se, ref["mean"], ref["se"])
    assert abs(r["logzerr"].mean() - ref["mean_logzerr"]) < 0.005
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.03
    if ref["K"] == K:
        assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.08


def test_c4_queue_size_effect_is_the_references_own(ctx):
    """With a whole live set of proposals in flight (K = nlive = 4000) ln Z comes out ~0.4 lower than serially -- in
    the real reference (two runs, SerialPool(4000), 5 000 s each) as on the device: the device run at K = 4000 is
    held to the reference's runs AT K = 4000, and the reference's own shift K = 1 -> 4000 is checked to be there."""
    from dynesty_amd import backend, nested, problems
    ref = json.load(open(os.path.join(GOLD, "c4_logz_ref.json")))
    k1 = np.array([x["logz"] for x in ref["runs"] if x["K"] == 1])
    k4 = [x for x in ref["runs"] if x["K"] == 4000]
    z4 = np.array([x["logz"] for x in k4])
    err = float(np.mean([x["logzerr"] for x in k4]))
    assert len(z4) >= 2 and k1.mean() - z4.mean() > 2.0 * err  # the reference's own queue-size effect
    prob = problems.gauss_normal_prior(200, "C4")
    backend.set_backend(ctx)
    try:
        r = nested.run_static(prob, nlive=4000, bound='single', sample='rslice', slices=203, queue_size=4000,
                              rstate=np.random.default_rng(21), dlogz=0.01)
    finally:
        backend.set_backend(None)
    assert abs(r.logz - z4.mean()) < 3.0 * err * math.sqrt(1 + 1 / len(z4)), (r.logz, z4)
    assert abs(r.niter / np.mean([x["niter"] for x in k4]) - 1) < 0.05


@pytest.mark.parametrize("bnd,rng", [("single", "pcg64"), ("multi", "pcg64"), ("single", "philox")])
def test_c1_device_unif_ensemble_vs_reference_ensemble(ctx, bnd, rng):
    """BASELINE config C1 with the reference's defaults for sample='unif' (bootstrap 5, enlarge 1) through the
    device-resident loop, against 32 runs of the real reference at the same queue size (K = 64) and serial."""
    g = json.load(open(os.path.join(GOLD, "c1_logz_ref.json")))["ensembles"]
    prob = inputs.problem("C1")
    r = ctx.ns_ensemble(prob, 64, 500, 64, bound=bnd, sample="unif", entropy=[2026, 1], dlogz=0.01, rng=rng)
    assert np.all(r["status"] == 0)
    lz = r["logz"]
    mean, se = lz.mean(), lz.std(ddof=1) / math.sqrt(len(lz))
    for key in (f"{bnd}_K64", f"{bnd}_K1"):
        ref = g[key]
        assert abs(mean - ref["mean"]) < bound(se, ref["se"]), (key, mean, se, ref["mean"], ref["se"])
    ref = g[f"{bnd}_K64"]
    assert 0.6 < lz.std(ddof=1) / ref["std"] < 1.6
    assert abs(r["logzerr"].mean() - ref["mean_logzerr"]) < 0.01
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.02
    # calls and bounds per run: the bootstrap-expanded bound is as tight as the reference's
    assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.08
    assert abs(r["nbound"].mean() / ref["mean_nbound"] - 1) < 0.25


@pytest.mark.parametrize("K,bound,ref_key", [(1, "multi", "multi_K1"), (64, "multi", "multi_K64"), (64, "single", "single_K64")])
def test_forced_bound_updates_above_the_register_dimensions(ctx, K, bound, ref_key):
    """Sampler.propose_live rebuilds the bound when a walker's start point lies outside it (sampler.py:484-489).
    At 40 dimensions with 333 live points that happens all the time -- the ellipsoid of so few points, enlarged by
    1.25 in VOLUME (0.6 % in radius), misses most new live points: the reference makes ~120 bound updates where the
    call-count schedule alone makes ~45 -- and with an under-mixed rwalk (walks = 60) the evidence depends on it by
    several nats at queue size 64 (the real reference: -85.7 +- 0.4 multi / -80.6 +- 0.6 single; without the forced
    updates the device loop gave -83.2 / -76.3; tests/golden/c40_logz_ref.json by tools/ref_c40_runs.py).  The check
    was not built above 32 dimensions until the shape sweep of tools/fuzz_loop.py met this case."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "c40_logz_ref.json")))["groups"][ref_key]
    prob = problems.gauss_corr(40, 0.3, 5.0, "c40")
    runs = 16
    r = ctx.ns_ensemble(prob, runs, 333, K, bound=bound, sample="rwalk", walks=60, dlogz=0.5, entropy=[K, 40])
    assert (r["status"] == 0).all()
    lz = r["logz"]
    se = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 3.5 * se, (lz.mean(), ref["mean"], se)
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.03
    # bound updates: well above the call-count schedule's ncall / (walks * nlive)
    scheduled = r["ncall"].mean() / (60 * 333)
    assert r["nbound"].mean() > 1.5 * scheduled, (r["nbound"].mean(), scheduled)


SHAPES = ["rslice40_multi", "slice3_Kgtn", "unif5_multi", "rwalk25_Kgtn", "rslice_egg", "rwalk13_multi", "unif_egg", "rwalk_egg",
          "slice9_single", "rwalk25_K1", "multi2_tiny", "unif3_Kgtn", "slice36_single", "rwalk64_multi",
          "rslice48_boot3", "rwalk44_multi", "unif9_single_enl", "slice2_egg"]


@pytest.mark.parametrize("case", SHAPES)
@pytest.mark.parametrize("rng", ["pcg64", "philox"])
def test_odd_shapes_vs_reference_ensembles(ctx, case, rng):
    """Shapes away from the BASELINE configs, each against an ensemble of the REAL reference at the same settings
    (tests/golden/shape_logz_ref.json, tools/ref_shape_runs.py / shape_cases.json): rslice in 40 dimensions with 320
    live points (wide walkers, narrow multi-ellipsoid rebuild, forced updates), queues LARGER than the live set
    (slice 3-D nlive 60 K 257; rwalk 25-D nlive 60 K 257, whose ln Z is 5 nats off the truth in the reference and
    here alike), unif + bootstrap on a correlated 5-D problem with the multi bound, small eggbox runs by rslice / unif (the
    reference's own default for 2-D) / rwalk, rwalk 13-D multi, slice 9-D single at K = 7, rwalk 25-D at K = 1, 25 live points in 2-D, unif with K > nlive, slice 36-D (wide
    walkers) and rwalk 64-D with the multi bound (the host-driven wide MultiEllipsoid.update inside the loop), rslice 48-D
    with three bootstrap replicas (the ragged wide batch inside the loop), rwalk 44-D multi (the last narrow rebuild
    dimension), unif 9-D with enlarge 1.5 and no bootstrap, slice on the eggbox.  Ensemble ln Z within 4 combined standard errors, iterations within 3 %, calls within 6 %."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge") if k in c}
    runs = 16 if c["prob"][1] >= 36 else 48
    r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                        entropy=[11, len(case)], rng=rng, **kw)
    assert (r["status"] == 0).all()
    lz = r["logz"]
    se = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 4.0 * se, (lz.mean(), ref["mean"], se)
    tol = 0.03 if ref["n"] >= 30 else 0.05  # (the small reference ensembles of the expensive shapes)
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < tol, (r["niter"].mean(), ref["mean_niter"])
    assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 2 * tol, (r["ncall"].mean(), ref["mean_ncall"])


OPTION_SHAPES = ["opt_update_interval", "opt_first_update", "opt_maxiter", "opt_maxcall", "opt_add_live", "opt_logl_max",
                 # a shape the loop fuzzer (tools/fuzz_loop.py) flagged against the ANALYTIC ln Z: 32-D, a queue of more
                 # than a third of the live set, a bound update every fill -- the reference's own scatter there is 1.2
                 # nats per run (three times its error estimate); 16 reference runs
                 "fuzz_g32_upd"]


@pytest.mark.parametrize("case", OPTION_SHAPES)
def test_run_loop_options_vs_reference_ensembles(ctx, case):
    """the options the reference takes in NestedSampler(update_interval=, first_update=) and
    run_nested(maxiter=, maxcall=, logl_max=, add_live=) (dynesty.py:213-234, sampler.py:1070-1093, 1214-1341), fixed
    at their defaults in the resident loop until round 4, each with a non-default value against an ensemble of 32
    REAL reference runs at the same settings (tools/shape_cases.json, tools/ref_shape_runs.py,
    tests/golden/shape_logz_ref.json): a float update_interval of 0.6 nlive calls (a bound update every few fills
    instead of every 33 nlive calls), a late first update (min_ncall 900, min_eff 30 %), maxiter = 700 (exactly 701
    deaths, as the reference's loop counter has it), maxcall = 30 000, add_live = False (the record is the dead
    points' running evidence), logl_max = -4."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge", "update_interval", "first_update", "maxiter",
                            "maxcall", "logl_max", "add_live") if k in c}
    runs = 64
    r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                        entropy=[13, len(case)], **kw)
    assert (r["status"] == 0).all()
    lz = r["logz"]
    se = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 4.0 * se, (lz.mean(), ref["mean"], se)
    if case == "opt_maxiter":
        assert (r["niter"] == c["maxiter"] + 1).all(), r["niter"]
    else:
        assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.04, (r["niter"].mean(), ref["mean_niter"])
    assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.08, (r["ncall"].mean(), ref["mean_ncall"])
    if case == "opt_maxcall":
        # the loop stops at the first death after which the calls exceed maxcall: never more than one fill beyond it
        assert (r["ncall"] > c["maxcall"] * 0.9).all() and (r["ncall"] < c["maxcall"] + 40 * c["K"] * 10).all()
    if case in ("opt_update_interval", "opt_first_update"):
        # bound updates follow the option (the reference counts its initial unit-cube bound as one)
        assert abs(r["nbound"].mean() / (ref["mean_nbound"] - 1) - 1) < 0.15, (r["nbound"].mean(), ref["mean_nbound"])


@pytest.mark.parametrize("case", ["rwalk44_multi", "rslice40_multi", "slice36_single"])
def test_forced_update_inside_the_fill_reproduces_the_reference_bound_counts(ctx, case):
    """forced_exact=True (DH_NS_OPT_FORCED_EXACT) takes propose_live's forced bound update
    (sampler.py:484-489) inside the fill that finds the start point outside the bound -- the reference's sequence for
    any queue size -- instead of flagging the run for its next fill.  On the shapes where forced updates are a large
    share of all updates (40-D class problems with ~8 live points per dimension) the number of bound updates per run
    then agrees with the ensembles of REAL reference runs to 2 % (the reference counts its initial unit-cube bound as
    one; the default form, whose forced updates mostly coincide with the regular update of the next fill, makes up to
    20 % fewer), and ln Z agrees within the ensembles' errors in both forms."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge") if k in c}
    runs = 64
    out = {}
    for exact in (False, True):
        r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"],
                            dlogz=c.get("dlogz", 0.5), entropy=[31, 7], forced_exact=exact, **kw)
        assert (r["status"] == 0).all()
        lz = r["logz"]
        se = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
        assert abs(lz.mean() - ref["mean"]) < 4.0 * se, (exact, lz.mean(), ref["mean"], se)
        out[exact] = r
    nb_ref = ref["mean_nbound"] - 1.0
    assert abs(out[True]["nbound"].mean() / nb_ref - 1) < 0.02, (out[True]["nbound"].mean(), nb_ref)
    assert out[False]["nbound"].mean() < out[True]["nbound"].mean()


@pytest.mark.parametrize("case", ["rwalk25_K1", "rwalk13_multi", "multi2_tiny"])
def test_bound_update_counts_of_the_default_protocol_vs_reference_ensembles(ctx, case):
    """the other three shape cases of profiles/r05/forced_exact_forms.jsonl, in the default
    (the reference's) protocol.  Counting conventions: the reference's `Sampler.nbound` starts at 1 -- its initial
    UnitCube is bound number one (sampler.py:416) and every update adds one (:673) -- while the loop's record counts
    the updates, so nbound(loop) is held to nbound(reference) - 1: that is the whole of round 5's "exactly -1.0" of
    `multi2_tiny` (0.875 updates on both sides) and of `rwalk13_multi` (15.98 against 15.94).  Gate: 2 % or 2 standard
    errors of the two ensembles' means (the per-run scatter of the count is about its square root), whichever is larger."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "slices", "bootstrap", "enlarge") if k in c}
    runs = 64
    r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                        entropy=[31, 7], **kw)
    assert (r["status"] == 0).all()
    nb, nb_ref = r["nbound"].astype(np.float64), ref["mean_nbound"] - 1.0
    se = math.hypot(nb.std(ddof=1) / math.sqrt(runs), math.sqrt(max(nb_ref, 1.0)) / math.sqrt(ref["n"]))
    assert abs(nb.mean() - nb_ref) < max(0.02 * nb_ref, 2.0 * se), (nb.mean(), nb_ref, se)
    lz = r["logz"]
    sez = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 4.0 * sez, (lz.mean(), ref["mean"], sez)


BC_SHAPES = ["bc_reflect_egg", "bc_mixed_egg", "bc_periodic_unif_egg"]


@pytest.mark.parametrize("case", BC_SHAPES)
def test_periodic_and_reflective_coordinates_in_the_resident_loop(ctx, case):
    """NestedSampler(periodic=, reflective=) reach the internal samplers only (dynesty.py:126-142): rwalk wraps /
    reflects the flagged coordinates of a proposal and tests them against (-0.5, 1.5) (internal_samplers.py:1023-1032),
    the uniform sampler widens its unitcheck and hands the unwrapped point on (:301-314).  The eggbox has modes ON
    the faces of the cube, so the flags change which proposals survive there.  Against ensembles of 32 REAL reference
    runs with the same flags (tools/shape_cases.json; the reference needs at least one hard coordinate: its unitcheck
    takes the minimum over them): 2-D reflective, 3-D periodic + reflective + hard, 2-D periodic with the uniform
    sampler."""
    from dynesty_amd import problems
    ref = json.load(open(os.path.join(GOLD, "shape_logz_ref.json")))["cases"][case]
    c = ref["config"]
    prob = getattr(problems, c["prob"][0])(*c["prob"][1:])
    kw = {k: c[k] for k in ("walks", "periodic", "reflective") if k in c}
    # (the uniform case: the median of a bimodal call count over 64 runs scatters by ~4 % -- 27.8 k against the
    # reference's 31.4 k once in round 5, 30.9 - 31.3 k in four ensembles of 256: 256 runs there)
    runs = 256 if case == "bc_periodic_unif_egg" else 64
    r = ctx.ns_ensemble(prob, runs, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                        entropy=[17, len(case)], **kw)
    assert (r["status"] == 0).all()
    lz = r["logz"]
    se = math.hypot(lz.std(ddof=1) / math.sqrt(runs), ref["se"])
    assert abs(lz.mean() - ref["mean"]) < 4.0 * se, (lz.mean(), ref["mean"], se)
    assert abs(r["niter"].mean() / ref["mean_niter"] - 1) < 0.04, (r["niter"].mean(), ref["mean_niter"])
    if case == "bc_periodic_unif_egg":
        # one run in five spends ten times the calls of the others (an ellipsoid that reaches across the widened face
        # of the cube: most draws fail the unitcheck) -- in the reference (6 of 32 runs) as on the device: medians and
        # the share of such runs are compared, not the means of a bimodal distribution
        rc = np.array(ref["ncalls"])
        assert abs(np.median(r["ncall"]) / np.median(rc) - 1) < 0.10, (np.median(r["ncall"]), np.median(rc))
        fd, fr = (r["ncall"] > 1e5).mean(), (rc > 1e5).mean()
        assert abs(fd - fr) < 3.0 * math.sqrt(fr * (1 - fr) * (1 / runs + 1 / len(rc))), (fd, fr)
    else:
        assert abs(r["ncall"].mean() / ref["mean_ncall"] - 1) < 0.10, (r["ncall"].mean(), ref["mean_ncall"])
    # the flags do something: the same ensemble without them proposes differently
    plain = ctx.ns_ensemble(prob, 8, c["nlive"], c["K"], bound=c["bound"], sample=c["sample"], dlogz=c.get("dlogz", 0.5),
                            entropy=[17, len(case)], **{k: v for k, v in kw.items() if k == "walks"})
    assert (plain["ncall"] != r["ncall"][:8]).any()
