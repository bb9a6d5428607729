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
