"""Drop-in boundary exercised with the REAL dynesty and
the oracle test backend on CPU: the plugin classes must be accepted by an
unmodified NestedSampler, survive deepcopy / pickling, honour dynesty's
isinstance-keyed defaults, and the batching pool must turn a queue fill into
one runner call.  (The same bound classes and samplers run on the HIP backend
in tests/test_gpu_endtoend.py / test_gpu_friends.py through the dynesty-free
driver, dynesty not being installed on the GPU box.)

Model: reference tests/test_bound_interface.py and test_sampler_interface.py.
"""
import copy
import pickle

import numpy as np
import pytest

import refshim

pytestmark = [
    pytest.mark.reference,
    pytest.mark.skipif(not refshim.have_reference(),
                       reason="needs /root/reference (build container)"),
]


@pytest.fixture(scope="module")
def dyn():
    d = refshim.import_reference()
    from dynesty_amd import backend
    from oracle_backend import OracleBackend
    backend.set_backend(OracleBackend())
    yield d
    backend.set_backend(None)


def run(dyn, prob, bound, sample, pool=None, nlive=200, seed=5, dlogz=0.5,
        **kw):
    s = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim,
                          nlive=nlive, bound=bound, sample=sample, pool=pool,
                          queue_size=None if pool is None else pool.size,
                          rstate=np.random.default_rng(seed), **kw)
    s.run_nested(dlogz=dlogz, print_progress=False)
    return s


def test_classes_are_reference_subclasses(dyn):
    from dynesty import bounding as db, internal_samplers as dis
    from dynesty_amd import dropin
    assert issubclass(dropin.HipMultiEllipsoid, db.Bound)
    assert issubclass(dropin.HipEllipsoid, db.Bound)
    assert issubclass(dropin.HipRWalkSampler, dis.RWalkSampler)
    assert issubclass(dropin.HipRSliceSampler, dis.RSliceSampler)
    assert issubclass(dropin.HipSliceSampler, dis.SliceSampler)
    assert issubclass(dropin.HipUniformBoundSampler, dis.UniformBoundSampler)


def test_unif_single_c1(dyn):
    """BASELINE config C1 shape: 3-D Gaussian, single bound, unif sampler."""
    import inputs
    from dynesty_amd import dropin
    prob = inputs.problem("C1")
    s = run(dyn, prob, dropin.HipEllipsoid(3),
            dropin.HipUniformBoundSampler(problem=prob), nlive=300)
    r = s.results
    assert abs(r.logz[-1] - prob.logz_truth) < 5 * r.logzerr[-1] + 0.1
    # dynesty keys bootstrap=5 / enlarge=1 on isinstance(UniformBoundSampler)
    assert s.bound_bootstrap == 5 and s.bound_enlarge == 1


@pytest.mark.parametrize("which", ["rwalk", "rslice", "slice"])
def test_multi_samplers_g5(dyn, which):
    import inputs
    from dynesty_amd import dropin
    prob = inputs.problem("G5")
    smp = dict(rwalk=dropin.HipRWalkSampler(problem=prob, walks=20),
               rslice=dropin.HipRSliceSampler(problem=prob, slices=4),
               slice=dropin.HipSliceSampler(problem=prob, slices=2))[which]
    s = run(dyn, prob, dropin.HipMultiEllipsoid(5), smp, nlive=150, dlogz=1.0)
    r = s.results
    assert abs(r.logz[-1] - prob.logz_truth) < 5 * r.logzerr[-1] + 0.3
    assert s.bound_enlarge == 1.25 and s.bound_bootstrap == 0
    assert isinstance(s.bound, dropin.HipMultiEllipsoid)
    assert s.nbound > 1


def test_batch_pool_one_launch_per_fill(dyn):
    import inputs
    from dynesty_amd import dropin, samplers
    prob = inputs.problem("G5")
    calls = []
    orig = samplers.run_rwalk

    class CountingPool(dropin.HipBatchPool):

        def map(self, func, iterable):
            items = list(iterable)
            if getattr(func, '_dynhip_batch', None) is not None:
                calls.append(len(items))
            return super().map(func, items)

    pool = CountingPool(queue_size=32)
    s = run(dyn, prob, dropin.HipMultiEllipsoid(5),
            dropin.HipRWalkSampler(problem=prob, walks=15), pool=pool,
            nlive=120, dlogz=2.0)
    assert calls and all(c == 32 for c in calls)
    r = s.results
    assert abs(r.logz[-1] - prob.logz_truth) < 5 * r.logzerr[-1] + 0.5
    assert samplers.run_rwalk is orig


def test_deepcopy_pickle_roundtrip(dyn):
    import inputs
    from dynesty_amd import dropin
    b = dropin.HipMultiEllipsoid(5)
    b.update(inputs.cloud("two5"), rstate=np.random.default_rng(1))
    assert b.nells == 2
    for clone in (copy.deepcopy(b), pickle.loads(pickle.dumps(b))):
        assert clone.nells == b.nells
        np.testing.assert_array_equal(clone.ams, b.ams)
        x = inputs.cloud("two5")[0]
        assert clone.contains(x) == b.contains(x) is True
        np.testing.assert_array_equal(
            clone.samples(5, rstate=np.random.default_rng(2)),
            b.samples(5, rstate=np.random.default_rng(2)))
    smp = dropin.HipRWalkSampler(problem=inputs.problem("G5"), walks=30)
    smp2 = pickle.loads(pickle.dumps(smp))
    assert smp2.sampler_kwargs['walks'] == 30
    assert smp2.sampler_kwargs['problem'].ndim == 5
    # dynesty re-instantiates the user's sampler from a template
    smp3 = smp._new_from_template(dict(ndim=5, ncdim=5, nonbounded=None,
                                       periodic=None, reflective=None,
                                       facc=0.5))
    assert type(smp3) is type(smp)
    assert smp3.sampler_kwargs['problem'] is smp.sampler_kwargs['problem']


def test_checkpoint_resume(dyn, tmp_path):
    """utils.save_sampler / restore_sampler with plugin objects inside."""
    import inputs
    from dynesty_amd import dropin
    prob = inputs.problem("G5")
    s = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, 5,
                          nlive=100, bound=dropin.HipMultiEllipsoid(5),
                          sample=dropin.HipRWalkSampler(problem=prob, walks=15),
                          rstate=np.random.default_rng(3))
    s.run_nested(maxiter=300, print_progress=False, add_live=False)
    f = str(tmp_path / "ck.sav")
    s.save(f)
    s2 = dyn.NestedSampler.restore(f)
    s2.run_nested(dlogz=2.0, print_progress=False, resume=True)
    assert abs(s2.results.logz[-1] - prob.logz_truth) < 1.5


def test_bound_api_matches_reference_semantics(dyn):
    """Method-by-method comparison with the reference classes on the same
    points (exact where the arithmetic is identical)."""
    import inputs
    from dynesty import bounding as db
    from dynesty_amd import dropin
    pts = inputs.cloud("two5")
    ours, ref = dropin.HipMultiEllipsoid(5), db.MultiEllipsoid(5)
    np.testing.assert_allclose(ours.logvol, ref.logvol, rtol=1e-14)
    ours.update(pts, rstate=np.random.default_rng(1))
    ref.update(pts, rstate=np.random.default_rng(1))
    assert ours.nells == ref.nells == 2
    np.testing.assert_allclose(ours.logvol, ref.logvol, rtol=1e-13)
    ours.scale_to_logvol(ours.logvol + np.log(1.25))
    ref.scale_to_logvol(ref.logvol + np.log(1.25))
    np.testing.assert_allclose(np.sort(ours.logvol_ells),
                               np.sort(ref.logvol_ells), rtol=1e-13)
    x = pts[3]
    assert ours.contains(x) == ref.contains(x)
    assert list(ours.within(x)) == list(ref.within(x)) or ours.nells == 2
    assert ours.overlap(x) == ref.overlap(x)
    single, rs = dropin.HipEllipsoid(5), db.Ellipsoid(5)
    single.update(pts, rstate=np.random.default_rng(1))
    rs.update(pts, rstate=np.random.default_rng(1))
    np.testing.assert_allclose(single.logvol, rs.logvol, rtol=1e-13)
    np.testing.assert_allclose(single.distance(x), rs.distance(x), rtol=1e-12)
    with pytest.raises(RuntimeError):
        ours.update(pts[:1], rstate=np.random.default_rng(1))


def test_rwalk_arbitrary_python_likelihood(dyn):
    """No device twin: the lock-step path proposes on the device (oracle
    backend here) and evaluates the user's Python callbacks on the host."""
    from dynesty_amd import dropin

    def loglike(v):
        return -0.5 * float(np.sum(v**2)) - 0.5 * 3 * np.log(2 * np.pi)

    def ptform(u):
        return 10.0 * (2.0 * u - 1.0)

    s = dyn.NestedSampler(loglike, ptform, 3, nlive=150,
                          bound=dropin.HipMultiEllipsoid(3),
                          sample=dropin.HipRWalkSampler(walks=15),
                          pool=dropin.HipBatchPool(16), queue_size=16,
                          rstate=np.random.default_rng(9))
    s.run_nested(dlogz=1.0, print_progress=False)
    r = s.results
    assert abs(r.logz[-1] - (-3 * np.log(20.0))) < 5 * r.logzerr[-1] + 0.4


@pytest.mark.parametrize("kind", ["balls", "cubes"])
def test_friends_bounds_dropin(dyn, kind):
    """bound=HipRadFriends / HipSupFriends in an unmodified NestedSampler, with
    the batched uniform sampler (1/q rule over the live-point shapes) and with
    rwalk (get_random_axes = the common shape)."""
    import inputs
    from dynesty import bounding as db
    from dynesty_amd import dropin
    prob = inputs.problem("C1")
    cls = dict(balls=dropin.HipRadFriends, cubes=dropin.HipSupFriends)[kind]
    assert issubclass(cls, db.Bound)
    pool = dropin.HipBatchPool(queue_size=8)
    s = run(dyn, prob, cls(3), dropin.HipUniformBoundSampler(problem=prob), pool=pool, nlive=120, dlogz=0.5)
    r = s.results
    assert abs(r.logz[-1] - prob.logz_truth) < 5 * r.logzerr[-1] + 0.15
    assert isinstance(s.bound, cls) and s.nbound > 1
    assert s.bound_bootstrap == 5  # dynesty's default for the uniform sampler
    s2 = run(dyn, prob, cls(3), dropin.HipRWalkSampler(problem=prob, walks=15), nlive=100, dlogz=1.0)
    r2 = s2.results
    assert abs(r2.logz[-1] - prob.logz_truth) < 5 * r2.logzerr[-1] + 0.3


@pytest.mark.parametrize("kind", ["balls", "cubes"])
def test_friends_api_matches_reference(dyn, kind):
    """Same attributes / method results as the reference classes on the same
    inputs (update, scale_to_logvol, within / overlap / contains, same-seed
    sample(s), monte_carlo_logvol, pickling)."""
    import inputs
    from dynesty import bounding as db
    from dynesty_amd import dropin
    pts = inputs.cloud("two5")
    ours = dict(balls=dropin.HipRadFriends, cubes=dropin.HipSupFriends)[kind](5)
    ref = dict(balls=db.RadFriends, cubes=db.SupFriends)[kind](5)
    for b in (ours, ref):
        b.update(pts, rstate=np.random.default_rng(1), bootstrap=0)
        b.update(pts, rstate=np.random.default_rng(1), bootstrap=2)
        b.scale_to_logvol(b.logvol + np.log(1.3))
        b.ctrs = pts
    for k in ("cov", "am", "axes", "axes_inv"):
        np.testing.assert_allclose(getattr(ours, k), np.real(getattr(ref, k)), rtol=0,
                                   atol=1e-12 * np.abs(getattr(ref, k)).max())
    assert abs(ours.logvol - ref.logvol) < 1e-10
    x = pts[3] + 0.01
    np.testing.assert_array_equal(ours.within(x), ref.within(x))
    assert ours.overlap(x) == ref.overlap(x) and ours.contains(x) == ref.contains(x)
    assert not ours.contains(np.full(5, 5.0))
    a, b = np.random.default_rng(3), np.random.default_rng(3)
    np.testing.assert_allclose(ours.samples(5, rstate=a), ref.samples(5, rstate=b), atol=1e-13)
    xa, qa = ours.sample(rstate=a, return_q=True)
    xb, qb = ref.sample(rstate=b, return_q=True)
    np.testing.assert_allclose(xa, xb, atol=1e-13)
    assert qa == qb and a.random() == b.random()  # generators left in the same state
    la, fa = ours.monte_carlo_logvol(200, rstate=np.random.default_rng(4))
    lb, fb = ref.monte_carlo_logvol(200, rstate=np.random.default_rng(4))
    assert abs(la - lb) < 1e-10 and abs(fa - fb) < 1e-12
    assert ours.get_random_axes(a) is ours.axes
    c = pickle.loads(pickle.dumps(copy.deepcopy(ours)))
    np.testing.assert_array_equal(c.axes_inv, ours.axes_inv)
    assert c.need_centers and c.kind == kind


def test_dynamic_nested_sampler_dropin(dyn):
    """The same plugin objects under an unmodified DynamicNestedSampler (baseline
    run + one posterior-weighted batch): bound re-instantiation from the template
    (_new_from_template), batch-wise bound updates and the merged result."""
    import inputs
    from dynesty_amd import dropin
    prob = inputs.problem("G5")
    pool = dropin.HipBatchPool(queue_size=8)
    s = dyn.DynamicNestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim,
                                 bound=dropin.HipMultiEllipsoid(5),
                                 sample=dropin.HipRWalkSampler(problem=prob, walks=15),
                                 pool=pool, queue_size=pool.size,
                                 rstate=np.random.default_rng(11))
    s.run_nested(nlive_init=100, nlive_batch=50, maxbatch=1, dlogz_init=0.5, print_progress=False)
    r = s.results
    assert abs(r.logz[-1] - prob.logz_truth) < 5 * r.logzerr[-1] + 0.3
    assert len(np.unique(r.samples_batch)) == 2  # baseline + one batch
    assert isinstance(s.sampler.bound, dropin.HipMultiEllipsoid)
    # posterior moments from the weighted samples: unit variance, zero mean
    w = r.importance_weights()
    mean = (w[:, None] * r.samples).sum(0)
    assert np.abs(mean).max() < 0.3


@pytest.mark.parametrize("bound", ["multi", "balls"])
def test_unif_arbitrary_python_likelihood(dyn, bound):
    """HipUniformBoundSampler() without a device problem: candidates come from
    the backend in lock step, the user's Python callbacks decide -- the default
    dynesty configuration (bound='multi', sample='unif') as a drop-in."""
    from dynesty_amd import dropin
    calls = []

    def loglike(v):
        calls.append(1)
        return -0.5 * float(np.sum(v**2)) - 1.5 * np.log(2 * np.pi)

    def ptform(u):
        return 10. * (2. * u - 1.)
    b = dropin.HipMultiEllipsoid(3) if bound == "multi" else dropin.HipRadFriends(3)
    s = dyn.NestedSampler(loglike, ptform, 3, nlive=150, bound=b,
                          sample=dropin.HipUniformBoundSampler(),
                          pool=dropin.HipBatchPool(queue_size=8), queue_size=8,
                          rstate=np.random.default_rng(3))
    s.run_nested(dlogz=0.5, print_progress=False)
    r = s.results
    truth = -3 * np.log(20.)
    assert abs(r.logz[-1] - truth) < 5 * r.logzerr[-1] + 0.15
    # every likelihood call went through the user's callback (the proposals of the last,
    # partly consumed queue fill are evaluated but not recorded)
    assert 0 <= len(calls) - int(np.sum(r.ncall)) <= 8 * 20


@pytest.mark.parametrize("which,doubling,periodic", [
    ("rslice", False, False), ("rslice", True, False), ("slice", False, False),
    ("slice", True, False), ("rslice", False, True), ("slice", False, True)])
def test_slice_lockstep_equals_reference_samplers(dyn, which, doubling, periodic):
    """Arbitrary Python likelihood with sample='rslice' / 'slice': the lock-step path (directions,
    axis orders and uniforms from the backend -- dh_slice_feed on the device, its NumPy twin
    here -- and generic_slice_step's state machine on the host) against the reference's own
    RSliceSampler.sample / SliceSampler.sample on the same arguments: same points, same
    likelihoods, same call / expansion / contraction counters, same final stream."""
    from dynesty_amd import samplers
    IS = dyn.internal_samplers
    nd = 4
    rng = np.random.default_rng(12)
    A = rng.standard_normal((nd, nd))
    prec = A @ A.T + nd * np.eye(nd)

    def loglike(v):
        d = v - 0.3
        return -0.5 * float(d @ prec @ d)

    def ptform(u):
        return 4.0 * u - 2.0
    axes = [0.15 * (np.eye(nd) + 0.3 * rng.standard_normal((nd, nd))) for _ in range(2)]
    us = 0.5 + 0.05 * rng.standard_normal((6, nd))
    lls = [loglike(ptform(x)) for x in us]
    loglstar = min(lls) - 0.5
    nonp = np.array([True, False, True, True]) if periodic else None
    kw = dict(slices=3, slice_doubling=doubling, nonperiodic=nonp)

    def mk(seed_objs, cls):
        return [cls(u=us[i].copy(), loglstar=loglstar, axes=axes[i % 2], scale=0.9,
                    prior_transform=ptform, loglikelihood=loglike, rseed=s, kwargs=dict(kw))
                for i, s in enumerate(seed_objs)]
    gens_a = [np.random.Generator(np.random.PCG64(100 + i)) for i in range(6)]
    gens_b = [np.random.Generator(np.random.PCG64(100 + i)) for i in range(6)]
    for g in gens_a + gens_b:
        g.integers(10)  # leave a buffered 32-bit half in the stream (has_uint32 = 1)
    ref_cls = IS.RSliceSampler if which == "rslice" else IS.SliceSampler
    ref = [ref_cls.sample(a) for a in mk(gens_a, IS.SamplerArgument)]
    run = samplers.run_rslice if which == "rslice" else samplers.run_slice
    got = run(mk(gens_b, IS.SamplerArgument))
    for r, g in zip(ref, got):
        np.testing.assert_array_equal(g.u, r.u)
        np.testing.assert_array_equal(g.v, r.v)
        assert g.logl == r.logl and g.ncalls == r.ncalls
        assert g.tuning_info == r.tuning_info
        assert g.proposal_stats == r.proposal_stats
    for a, b in zip(gens_a, gens_b):
        assert a.bit_generator.state == b.bit_generator.state


def test_rslice_arbitrary_python_likelihood_run(dyn):
    """A whole NestedSampler run with a Python likelihood and the drop-in rslice sampler."""
    from dynesty_amd import dropin

    def loglike(v):
        return -0.5 * float(np.sum(v**2)) - 1.5 * np.log(2 * np.pi)

    def ptform(u):
        return 10. * (2. * u - 1.)
    s = dyn.NestedSampler(loglike, ptform, 3, nlive=100, bound=dropin.HipMultiEllipsoid(3),
                          sample=dropin.HipRSliceSampler(), pool=dropin.HipBatchPool(queue_size=8),
                          queue_size=8, rstate=np.random.default_rng(4))
    s.run_nested(dlogz=0.5, print_progress=False)
    r = s.results
    assert abs(r.logz[-1] - (-3 * np.log(20.))) < 5 * r.logzerr[-1] + 0.2


def test_slice_lockstep_lookahead_refill(dyn, monkeypatch):
    """The uniform lookahead of the lock-step slice path running dry (here: 2 values per call) is
    refilled per walker without disturbing the stream: results still equal the reference's."""
    from dynesty_amd import samplers
    IS = dyn.internal_samplers
    orig = samplers._UniformFeed.__init__

    def tiny(self, be, ndim, states6, nlook):
        orig(self, be, ndim, states6, 2)
    monkeypatch.setattr(samplers._UniformFeed, "__init__", tiny)

    def loglike(v):
        return -0.5 * float(np.sum((v - 0.2)**2) / 0.3**2)

    def ptform(u):
        return 6.0 * u - 3.0
    rng = np.random.default_rng(5)
    us = 0.5 + 0.03 * rng.standard_normal((4, 3))
    loglstar = min(loglike(ptform(x)) for x in us) - 0.3
    axes = 0.2 * np.eye(3)
    for cls, run in ((IS.RSliceSampler, samplers.run_rslice), (IS.SliceSampler, samplers.run_slice)):
        def mk(gens):
            return [IS.SamplerArgument(u=us[i].copy(), loglstar=loglstar, axes=axes, scale=1.0,
                                       prior_transform=ptform, loglikelihood=loglike, rseed=g,
                                       kwargs=dict(slices=4, slice_doubling=False, nonperiodic=None))
                    for i, g in enumerate(gens)]
        ga = [np.random.Generator(np.random.PCG64(40 + i)) for i in range(4)]
        gb = [np.random.Generator(np.random.PCG64(40 + i)) for i in range(4)]
        ref = [cls.sample(a) for a in mk(ga)]
        got = run(mk(gb))
        for r, g in zip(ref, got):
            np.testing.assert_array_equal(g.u, r.u)
            assert g.logl == r.logl and g.ncalls == r.ncalls
        for a, b in zip(ga, gb):
            assert a.bit_generator.state == b.bit_generator.state


def test_evaluation_history_equals_the_reference_samplers(dyn):
    """save_evaluation_history=True (utils.LogLikelihood, utils.py:120-262) asks the samplers for every point they
    evaluated (internal_samplers.py:28-31, 311-339, 426, 663, 960-961, 1118-1119).  The fused kernels keep those in
    registers; a run that wants them takes the lock-step runners (the device proposes, the host evaluates and records:
    the opt-in slow path).  Held here to the reference's own samplers on the same generators:
    the same history, item for item, for rwalk, rslice, slice and the uniform sampler -- also with a device problem
    attached (which alone would select the fused kernel)."""
    import dynesty.internal_samplers as IS
    from dynesty import bounding as RB
    from dynesty_amd import samplers, problems
    prob = problems.gauss_iid(3, 10.0, "hist3")

    class LL:  # what utils.LogLikelihood looks like to a sampler
        save_evaluation_history = True

        def __call__(self, v):
            return prob.loglikelihood(v)
    rng = np.random.default_rng(8)
    us = rng.uniform(0.45, 0.55, size=(5, 3))
    loglstar = min(prob.loglikelihood(prob.prior_transform(u)) for u in us) - 0.5
    axes = 0.05 * np.eye(3)

    def same(ref, got):
        assert len(ref) == len(got)
        for r, g in zip(ref, got):
            np.testing.assert_array_equal(g.u, r.u)
            assert g.logl == r.logl and g.ncalls == r.ncalls
            assert len(g.evaluation_history) == len(r.evaluation_history) > 0
            for a, b in zip(r.evaluation_history, g.evaluation_history):
                np.testing.assert_array_equal(b.u, a.u)
                np.testing.assert_array_equal(b.v, a.v)
                assert b.logl == a.logl
    cases = [(IS.RWalkSampler, samplers.run_rwalk, dict(walks=12)),
             (IS.RSliceSampler, samplers.run_rslice, dict(slices=3, slice_doubling=False, nonperiodic=None)),
             (IS.SliceSampler, samplers.run_slice, dict(slices=2, slice_doubling=False, nonperiodic=None))]
    for cls, runner, kw in cases:
        for with_problem in (False, True):
            def mk(gens):
                k = dict(kw, problem=prob) if with_problem else dict(kw)
                return [IS.SamplerArgument(u=us[i].copy(), loglstar=loglstar, axes=axes, scale=1.0,
                                           prior_transform=prob.prior_transform, loglikelihood=LL(), rseed=g, kwargs=k)
                        for i, g in enumerate(gens)]
            ga = [np.random.Generator(np.random.PCG64(70 + i)) for i in range(5)]
            gb = [np.random.Generator(np.random.PCG64(70 + i)) for i in range(5)]
            same([cls.sample(a) for a in mk(ga)], samplers.batched(runner)._dynhip_batch(mk(gb)))
            for a, b in zip(ga, gb):
                assert a.bit_generator.state == b.bit_generator.state
    # the uniform sampler inside a bound
    bound = RB.Ellipsoid(3, ctr=np.full(3, 0.5), cov=0.01 * np.eye(3))
    for with_problem in (False, True):
        def mk(gens):
            k = dict(bound=bound, ndim=3, n_cluster=3, nonbounded=None)
            if with_problem:
                k["problem"] = prob
            return [IS.SamplerArgument(u=None, loglstar=loglstar, axes=None, scale=1.0, prior_transform=prob.prior_transform,
                                       loglikelihood=LL(), rseed=g, kwargs=k) for g in gens]
        ga = [np.random.Generator(np.random.PCG64(90 + i)) for i in range(5)]
        gb = [np.random.Generator(np.random.PCG64(90 + i)) for i in range(5)]
        same([IS.UniformBoundSampler.sample(a) for a in mk(ga)], samplers.batched(samplers.run_unif)._dynhip_batch(mk(gb)))


def test_pool_batches_dynestys_own_unit_cube_sampler(dyn):
    """The phase before the first bound is dynesty's own UnitCubeSampler (sampler.py builds it; no drop-in class can be
    handed in for it): HipBatchPool.map recognises its `sample` and runs the queue as one backend call when the
    callbacks are a device Problem's -- the same points, call counts and generator states as the serial map -- and
    leaves it alone otherwise (other callbacks, wrapped arguments, an evaluation history wanted)."""
    import dynesty.internal_samplers as IS
    from dynesty import utils as DU
    from dynesty_amd import backend, dropin, problems
    prob = problems.gauss_iid(3, 10.0, "cube3")
    pool = dropin.HipBatchPool(8)
    loglstar = -9.0
    kids = np.random.SeedSequence(4).spawn(8)

    def mk(ll, pt):
        return [IS.SamplerArgument(u=None, loglstar=loglstar, axes=None, scale=1.0, prior_transform=pt, loglikelihood=ll,
                                   rseed=k, kwargs=dict(ndim=3)) for k in kids]
    ll = DU.LogLikelihood(prob.loglikelihood, 3)
    calls = []
    be = backend.get_backend()
    orig = be.unif_batch

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)
    be.unif_batch = spy
    try:
        ref = [IS.UnitCubeSampler.sample(a) for a in mk(ll, prob.prior_transform)]
        got = pool.map(IS.UnitCubeSampler.sample, mk(ll, prob.prior_transform))
        assert len(calls) == 1
        for r, g in zip(ref, got):
            np.testing.assert_array_equal(g.u, r.u)
            np.testing.assert_allclose(g.v, r.v, rtol=0, atol=1e-12)
            assert g.ncalls == r.ncalls and abs(g.logl - r.logl.val if hasattr(r.logl, 'val') else g.logl - r.logl) < 1e-10
            assert g.tuning_info is None and g.proposal_stats == dict(n_proposals=r.ncalls)
        # not a device Problem's callbacks: the serial map
        got = pool.map(IS.UnitCubeSampler.sample, mk(lambda v: prob.loglikelihood(v), prob.prior_transform))
        assert len(calls) == 1 and len(got) == 8
    finally:
        be.unif_batch = orig
