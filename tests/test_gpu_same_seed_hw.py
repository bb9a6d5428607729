"""The same-seed whole-run comparison ON THE HIP BACKEND (north_star: "results must match
the pure-Python reference on identical RNG seeds"): the REAL dynesty run twice with the same seed --

  A  its own classes (bounding.py / internal_samplers.py on the host CPU), eigenvector signs fixed to the
     device's convention (tests/refshim.canonical_eigh);
  B  the drop-in classes (dropin.Hip*, HipBatchPool) on the default backend = libdynhip.so on cuda:0.

`sampler.py:1070-1212` drives `bound.update` / `sample` end to end in both; the runs must kill the same slots in
the same order under the same bounds with the same numbers of calls, ln L to 1e-10, ln Z to 1e-9.  What a device
run may differ in is rounding (ocml's exp / log, summation order of the frame product: 1e-13 in a coordinate), so
the first diverging iteration -- if there is one -- is written down together with how close the decision was.

dynesty is not installed on the GPU box and the reference tree does not travel: the test runs when
DYNESTY_REF_PY names a staged scratch copy (tools/stage_reference.sh; never committed) and skips otherwise.
Every run writes its record to gpurun_out/same_seed_hw.json (copied to profiles/r06/ by hand).
"""
import json
import os

import numpy as np
import pytest

import refshim
from test_same_seed_e2e import SerialPool, history

pytestmark = [
    pytest.mark.gpu,
    pytest.mark.reference,
    pytest.mark.skipif(not refshim.have_reference(), reason="needs a copy of the reference (DYNESTY_REF_PY)"),
]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "gpurun_out", "same_seed_hw.json")


def run_pair_hw(prob, nlive, bound, sample, K, seed, maxiter, drop_kw):
    dyn = refshim.import_reference()
    from dynesty_amd import backend, dropin
    with refshim.canonical_eigh():
        a = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive, bound=bound,
                              sample=sample, pool=SerialPool(K), queue_size=K,
                              rstate=np.random.default_rng(seed))
        a.run_nested(dlogz=0.01, maxiter=maxiter, print_progress=False)
    backend.set_backend(None)          # the default: the HIP context (raises without libdynhip.so / a GPU)
    be = backend.get_backend()
    assert "oracle" not in getattr(be, "name", type(be).__name__).lower()
    bnd, smp = drop_kw(dropin)
    b = dyn.NestedSampler(prob.loglikelihood, prob.prior_transform, prob.ndim, nlive=nlive, bound=bnd,
                          sample=smp, pool=dropin.HipBatchPool(queue_size=K), queue_size=K,
                          rstate=np.random.default_rng(seed))
    b.run_nested(dlogz=0.01, maxiter=maxiter, print_progress=False)
    return a, b


def compare(name, a, b):
    """The record of one pair: how far the two histories coincide, and what the first difference looks like."""
    ia, la, za, na, ba = history(a)
    ib, lb, zb, nb, bb = history(b)
    n = min(len(ia), len(ib))
    same = (ia[:n] == ib[:n]) & (na[:n] == nb[:n]) & (ba[:n] == bb[:n])
    first = int(np.argmin(same)) if not same.all() else None
    upto = n if first is None else first
    rec = dict(case=name, iterations_reference=int(len(ia)), iterations_hip=int(len(ib)), iterations_compared=int(n),
               identical_prefix=int(upto), first_diverging_iteration=first,
               max_rel_logl_diff_on_prefix=float(np.max(np.abs(la[:upto] - lb[:upto]) / np.maximum(1e-300, np.abs(la[:upto])))) if upto else None,
               logz_reference=float(a.results.logz[-1]), logz_hip=float(b.results.logz[-1]),
               logz_diff=float(b.results.logz[-1] - a.results.logz[-1]),
               ncall_reference=int(a.ncall), ncall_hip=int(b.ncall), nbound_reference=int(a.nbound), nbound_hip=int(b.nbound))
    if first is not None:
        j = first
        rec["divergence"] = dict(
            slot=(int(ia[j]), int(ib[j])), logl=(float(la[j]), float(lb[j])), ncalls=(int(na[j]), int(nb[j])),
            bounditer=(int(ba[j]), int(bb[j])),
            # how close the decision was: the gap between the two smallest live ln L around the divergence
            logl_gap_next=float(abs(la[j + 1] - la[j])) if j + 1 < len(la) else None)
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    try:
        allrec = json.load(open(RECORD))
    except Exception:
        allrec = {}
    allrec[name] = rec
    json.dump(allrec, open(RECORD, "w"), indent=1)
    return rec


def assert_same_run_hw(name, a, b, min_bounds):
    rec = compare(name, a, b)
    assert rec["first_diverging_iteration"] is None, rec
    ia, la, za, na, ba = history(a)
    ib, lb, zb, nb, bb = history(b)
    assert len(ia) == len(ib)
    np.testing.assert_allclose(la, lb, rtol=1e-10, atol=0)
    np.testing.assert_allclose(za[-1], zb[-1], rtol=0, atol=1e-9)
    assert a.ncall == b.ncall and a.it == b.it
    assert a.nbound == b.nbound >= min_bounds
    np.testing.assert_allclose(a.results.logz[-1], b.results.logz[-1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(a.results.samples_u, b.results.samples_u, rtol=0, atol=1e-10)


def test_c1_whole_run_same_seed_on_hip():
    """BASELINE C1 (3-D Gaussian, single bound, uniform sampler with 5 bootstrap replicas), the whole run."""
    import inputs
    prob = inputs.problem("C1")
    a, b = run_pair_hw(prob, 300, 'single', 'unif', 32, 2718, None,
                       lambda d: (d.HipEllipsoid(3), d.HipUniformBoundSampler(problem=prob)))
    assert_same_run_hw("C1_single_unif_nlive300_K32_whole_run", a, b, min_bounds=5)


def test_c2_short_same_seed_on_hip():
    """BASELINE C2 settings (25-D correlated Normal, multi / rwalk, walks 45) at nlive 400, K 64: ~10 bound updates."""
    import inputs
    prob = inputs.problem("C2")
    a, b = run_pair_hw(prob, 400, 'multi', 'rwalk', 64, 314, 3500,
                       lambda d: (d.HipMultiEllipsoid(25), d.HipRWalkSampler(problem=prob, walks=45)))
    assert_same_run_hw("C2_multi_rwalk_nlive400_K64_3500it", a, b, min_bounds=4)


def test_eggbox_rslice_same_seed_on_hip():
    """C3's shape (2-D eggbox, many ellipsoids, rslice), 2500 iterations."""
    import inputs
    prob = inputs.problem("C3")
    a, b = run_pair_hw(prob, 500, 'multi', 'rslice', 50, 99, 2500,
                       lambda d: (d.HipMultiEllipsoid(2), d.HipRSliceSampler(problem=prob, slices=5)))
    assert_same_run_hw("C3_eggbox_multi_rslice_nlive500_K50_2500it", a, b, min_bounds=3)
