"""Same-seed end-to-end harness (north_star: "results must match the pure-Python
reference on identical RNG seeds"): the REAL dynesty run twice with the same seed --

  A  its own classes (MultiEllipsoid / Ellipsoid, RWalkSampler / UniformBoundSampler, a serial pool), with
     the eigenvector signs of `lalg.eigh` fixed to the device's convention (tests/refshim.canonical_eigh;
     LAPACK's signs are arbitrary and decide the order of the k-means children);
  B  the drop-in classes (dropin.Hip*, HipBatchPool) on the oracle backend with the same sign convention.

The two runs must kill the same live-point slots in the same order, with the same log-likelihoods, and
end at the same ln Z: every random draw, every bound update, every accept decision coincides.  The HIP
backend is tied into the chain by tests/test_gpu_tapb_replay.py, which replays the backend calls recorded
from a run of kind B on the device and holds every return to the oracle backend's.
"""
import numpy as np
import pytest

import refshim

pytestmark = [
    pytest.mark.reference,
    pytest.mark.skipif(not refshim.have_reference(), reason="needs /root/reference (build container)"),
]


class SerialPool:
    def __init__(self, size):
        self.size = size

    def map(self, f, x):
        return list(map(f, x))


def history(s):
    sr = s.saved_run
    return (np.array(sr['id']), np.array(sr['logl'], dtype=np.float64), np.array(sr['logz'], dtype=np.float64),
            np.array(sr['nc']), np.array(sr['bounditer']))


def run_pair(prob, nlive, bound, sample, K, seed, maxiter, ref_kw, drop_kw):
    dyn = refshim.import_reference()
    from dynesty_amd import backend, dropin
    from oracle_backend import OracleBackend
    with refshim.canonical_eigh():
        a = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive, bound=bound,
                              sample=sample, pool=SerialPool(K), queue_size=K,
                              rstate=np.random.default_rng(seed), **ref_kw)
        a.run_nested(dlogz=0.01, maxiter=maxiter, print_progress=False)
    backend.set_backend(OracleBackend(canon=True))
    try:
        bnd, smp = drop_kw(dropin)
        b = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive, bound=bnd,
                              sample=smp, pool=dropin.HipBatchPool(queue_size=K), queue_size=K,
                              rstate=np.random.default_rng(seed))
        b.run_nested(dlogz=0.01, maxiter=maxiter, print_progress=False)
    finally:
        backend.set_backend(None)
    return a, b


def assert_same_run(a, b, min_bounds):
    ia, la, za, na, ba = history(a)
    ib, lb, zb, nb, bb = history(b)
    assert len(ia) == len(ib)
    np.testing.assert_array_equal(ia, ib)      # the same slot dies at every iteration
    np.testing.assert_array_equal(na, nb)      # with the same number of likelihood calls
    np.testing.assert_array_equal(ba, bb)      # under the same bound
    # positions / log-likelihoods agree to rounding, not bit for bit: the reference keeps its axes in
    # Fortran order (scipy eigh), the drop-in in C order, and NumPy's np.dot(axes, dr) then takes a
    # different BLAS kernel (summation order); the ~1e-16 per step random-walks to ~1e-12 over a run
    np.testing.assert_allclose(la, lb, rtol=1e-10, atol=0)
    np.testing.assert_allclose(za[-1], zb[-1], rtol=0, atol=1e-9)
    assert a.ncall == b.ncall and a.it == b.it
    assert a.nbound == b.nbound >= min_bounds
    np.testing.assert_allclose(a.results.logz[-1], b.results.logz[-1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(a.results.samples_u, b.results.samples_u, rtol=0, atol=1e-10)


def test_c2_short_same_seed():
    """BASELINE C2 settings (25-D, multi/rwalk) at nlive 400, K 64: ~10 bound updates."""
    import inputs
    prob = inputs.problem("C2")
    a, b = run_pair(prob, 400, 'multi', 'rwalk', 64, 314, 3500, {},
                    lambda d: (d.HipMultiEllipsoid(25), d.HipRWalkSampler(problem=prob, walks=45)))  # dynesty.py:128: 20 + ndim
    assert_same_run(a, b, min_bounds=4)


def test_c1_same_seed():
    """BASELINE C1 (3-D Gaussian, single bound, uniform sampler with 5 bootstrap replicas), whole run."""
    import inputs
    prob = inputs.problem("C1")
    a, b = run_pair(prob, 300, 'single', 'unif', 32, 2718, None, {},
                    lambda d: (d.HipEllipsoid(3), d.HipUniformBoundSampler(problem=prob)))
    assert_same_run(a, b, min_bounds=5)
    assert abs(a.results.logz[-1] - prob.logz_truth) < 5 * a.results.logzerr[-1] + 0.1


def test_eggbox_rslice_same_seed():
    """C3's shape (2-D eggbox, many ellipsoids, rslice), short."""
    import inputs
    prob = inputs.problem("C3")
    a, b = run_pair(prob, 500, 'multi', 'rslice', 50, 99, 2500, {},
                    lambda d: (d.HipMultiEllipsoid(2), d.HipRSliceSampler(problem=prob, slices=5)))  # 3 + ndim
    assert_same_run(a, b, min_bounds=3)
    assert max(x.nells for x in b.bound_list if hasattr(x, 'nells')) >= 4 if hasattr(b, 'bound_list') else True
