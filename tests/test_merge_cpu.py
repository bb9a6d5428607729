"""Ensemble combiner vs the REAL reference: `ensemble.merge_logz` must give the
same ln Z and error as dynesty.utils.merge_runs (utils.py:1817-1900) for static
runs of equal nlive (build container only: needs /root/reference)."""
import numpy as np
import pytest

import refshim

pytestmark = [
    pytest.mark.reference,
    pytest.mark.skipif(not refshim.have_reference(),
                       reason="needs /root/reference (build container)"),
]


def test_merge_logz_matches_reference_merge_runs():
    dynesty = refshim.import_reference()
    from dynesty import utils as dyu
    import inputs
    from dynesty_amd import ensemble
    prob = inputs.problem("C1")
    res, dead, live, nit = [], [], [], []
    for s in range(4):
        sm = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 3,
                                   nlive=100, bound='single', sample='unif',
                                   rstate=np.random.default_rng(s))
        sm.run_nested(dlogz=0.1, print_progress=False)
        r = sm.results
        res.append(r)
        dead.append(np.array(r.logl[:r.niter]))
        live.append(np.array(r.logl[r.niter:]))
        nit.append(r.niter)
    merged = dyu.merge_runs(res, print_progress=False)
    d = np.zeros((4, max(nit)))
    for i, x in enumerate(dead):
        d[i, :len(x)] = x
    lz, err = ensemble.merge_logz(d, nit, np.array(live))
    np.testing.assert_allclose(lz, merged.logz[-1], rtol=0, atol=1e-10)
    np.testing.assert_allclose(err, merged.logzerr[-1], rtol=1e-9)


def test_merge_static_runs_matches_reference_fields():
    """Every per-point field of the merged run (order, live counts, volumes,
    weights, cumulative ln Z / information / error, samples) equals the
    reference's utils.merge_runs on static runs."""
    dynesty = refshim.import_reference()
    from dynesty import utils as dyu
    import inputs
    from dynesty_amd import ensemble
    prob = inputs.problem("C1")
    R, N = 3, 60
    res = []
    for s in range(R):
        sm = dynesty.NestedSampler(prob.loglikelihood, prob.prior_transform, 3,
                                   nlive=N, bound='single', sample='unif',
                                   rstate=np.random.default_rng(100 + s))
        sm.run_nested(dlogz=0.3, print_progress=False)
        res.append(sm.results)
    ref = dyu.merge_runs(res, print_progress=False)
    nit = [r.niter for r in res]
    dead_l = np.zeros((R, max(nit)))
    dead_u = np.zeros((R, max(nit), 3))
    live_l = np.zeros((R, N))
    live_u = np.zeros((R, N, 3))
    dead_i = np.zeros((3, R, max(nit)), dtype=np.int64)
    live_it = np.zeros((R, N), dtype=np.int64)
    for i, r in enumerate(res):
        dead_l[i, :nit[i]] = r.logl[:nit[i]]
        dead_u[i, :nit[i]] = r.samples_u[:nit[i]]
        dead_i[0, i, :nit[i]] = r.samples_id[:nit[i]]
        dead_i[1, i, :nit[i]] = r.samples_it[:nit[i]]
        dead_i[2, i, :nit[i]] = r.ncall[:nit[i]]
        # the device hands the final live points over in SLOT order (slot = the reference's 'id'), not sorted
        sl = np.argsort(r.samples_id[nit[i]:])
        assert (np.asarray(r.samples_id[nit[i]:])[sl] == np.arange(N)).all()
        live_l[i] = np.asarray(r.logl[nit[i]:])[sl]
        live_u[i] = np.asarray(r.samples_u[nit[i]:])[sl]
        live_it[i] = np.asarray(r.samples_it[nit[i]:])[sl]

    def ptform(u):
        return np.array([prob.prior_transform(x) for x in u])
    m = ensemble.merge_static_runs(dead_l, nit, live_l, dead_u, live_u,
                                   prior_transform=ptform, dead_id=dead_i[0], dead_it=dead_i[1],
                                   dead_nc=dead_i[2], live_it=live_it)
    # the per-point bookkeeping _merge_two copies through (utils.py:2154-2156, 2196-2207)
    np.testing.assert_array_equal(m.samples_id, ref.samples_id)
    np.testing.assert_array_equal(m.samples_it, ref.samples_it)
    np.testing.assert_array_equal(m.ncall, ref.ncall)
    assert m.niter == ref.niter
    np.testing.assert_array_equal(m.logl, ref.logl)
    np.testing.assert_array_equal(m.samples_n, ref.samples_n)
    np.testing.assert_array_equal(m.samples_u, ref.samples_u)
    np.testing.assert_allclose(m.samples, ref.samples, rtol=0, atol=1e-14)
    np.testing.assert_allclose(m.logvol, ref.logvol, rtol=0, atol=1e-11)
    np.testing.assert_allclose(m.logwt, ref.logwt, rtol=0, atol=1e-10)
    np.testing.assert_allclose(m.logz, ref.logz, rtol=0, atol=1e-10)
    np.testing.assert_allclose(m.information, ref.information, rtol=0, atol=1e-9)
    np.testing.assert_allclose(m.logzerr, ref.logzerr, rtol=1e-7, atol=1e-10)
    w = m.importance_weights()
    np.testing.assert_allclose(w, ref.importance_weights(), rtol=1e-9, atol=1e-300)
    assert abs(m.eff - ref.eff) < 1e-9
