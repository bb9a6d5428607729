"""run_ensemble_device must not average failed / unconverged runs into the ensemble ln Z
(records carry the device loop's status)."""
import numpy as np
import pytest

from dynesty_amd import backend, ensemble, _lib


class FakeBackend:
    """Stands in for the HIP context: returns a canned dh_ns_ensemble result."""

    def __init__(self, status):
        self.status = np.asarray(status)

    def ns_ensemble(self, prob, runs, nlive, queue_size, **kw):
        assert runs == len(self.status)
        return dict(logz=np.linspace(-57.6, -57.4, runs), logzerr=np.full(runs, 0.1),
                    niter=np.full(runs, 80000), ncall=np.full(runs, 3500000),
                    h=np.full(runs, 28.0), status=self.status)


def test_failed_runs_raise_or_are_masked():
    backend.set_backend(FakeBackend([0, 0, -2, 1]))
    try:
        with pytest.raises(RuntimeError, match="status"):
            ensemble.run_ensemble_device(None, 4)
        table = ensemble.run_ensemble_device(None, 4, on_failure='nan')
        assert np.isnan(table[2:, 1]).all() and np.isfinite(table[:2, 1]).all()
        mean, se, n = ensemble.combine_logz(table)
        assert n == 2 and np.isclose(mean, table[:2, 1].mean())
        with pytest.raises(ValueError):
            ensemble.run_ensemble_device(None, 4, on_failure='ignore')
    finally:
        backend.set_backend(None)
    backend.set_backend(FakeBackend([0, 0, 0]))
    try:
        table = ensemble.run_ensemble_device(None, 3)
        assert ensemble.combine_logz(table)[2] == 3
    finally:
        backend.set_backend(None)


def test_ns_ensemble_rejects_unsupported_bound_and_sampler():
    """The device-resident loop knows 'multi' / 'single' and rwalk / rslice / slice / unif only; anything
    else must raise instead of silently running a single ellipsoid."""
    ctx = _lib.Context.__new__(_lib.Context)  # no device needed: validation comes first

    class P:
        ndim = 3
    for kw in (dict(bound='balls'), dict(bound='cubes'), dict(sample='hslice'), dict(bound='none')):
        with pytest.raises(ValueError, match="not supported"):
            _lib.Context.ns_ensemble(ctx, P(), 2, 100, 16, **kw)


def test_enlarge_bootstrap_defaults_follow_the_reference():
    """_get_enlarge_bootstrap (dynesty.py:169-200): unif -> (1, 5), others -> (1.25, 0); one given -> the other off;
    both only if one of them is neutral."""
    f = _lib.enlarge_bootstrap_defaults
    assert f('unif', None, None) == (1.0, 5)
    assert f('rwalk', None, None) == (1.25, 0) and f('rslice', None, None) == (1.25, 0)
    assert f('unif', 1.5, None) == (1.5, 0)
    assert f('rwalk', None, 7) == (1.0, 7) and f('rwalk', None, 0) == (1.0, 0)
    assert f('unif', 1.0, 5) == (1.0, 5) and f('unif', 1.3, 0) == (1.3, 0)
    with pytest.raises(ValueError):
        f('unif', 1.3, 5)
    with pytest.raises(ValueError):
        f('rwalk', None, 1)
    with pytest.raises(ValueError):
        f('rwalk', 0.9, None)


def test_run_static_refuses_an_nlive_the_device_cannot_hold():
    from dynesty_amd import nested

    class P:
        ndim = 3
    for nlive in (3, 65536, 200000):  # (round 4: up to 65535 live points, slots are 16-bit)
        with pytest.raises(ValueError, match="nlive"):
            nested.run_static(P(), nlive=nlive)
