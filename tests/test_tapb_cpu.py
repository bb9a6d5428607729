"""Record / replay machinery of the tap-B traces on CPU: the committed traces load, replaying them on the
oracle backend is deterministic, and the comparator does catch a wrong result."""
import os

import numpy as np
import pytest

import tapb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_trace_replay_on_oracle_backend_is_deterministic():
    from oracle_backend import OracleBackend
    meta, calls = tapb.load_trace(os.path.join(GOLD, "tapb_egg.npz"))
    assert meta["problem"] == "C3" and meta["calls"]["slice_batch"] >= 10
    calls = calls[:200]
    a = tapb.replay(calls, OracleBackend(canon=True))
    b = tapb.replay(calls, OracleBackend(canon=True))
    for (name, _, _), x, y in zip(calls, a, b):
        tapb.compare(name, x, y)
    # the comparator is not vacuous
    k = [i for i, c in enumerate(calls) if c[0] == "slice_batch"][0]
    bad = dict(a[k])
    bad["u"] = bad["u"] + 1e-9
    with pytest.raises(AssertionError):
        tapb.compare("slice_batch", bad, a[k])
    bad = dict(a[k])
    bad["ncalls"] = bad["ncalls"] + 1
    with pytest.raises(AssertionError):
        tapb.compare("slice_batch", bad, a[k])


def test_canonical_signs_fix_the_list_order():
    """With the device's sign convention the oracle's ellipsoid list has a defined order."""
    import inputs
    from oracle import bounding_ref as B
    pts = inputs.cloud("c3")
    old = B.CANON_SIGNS
    try:
        B.CANON_SIGNS = True
        m1 = B.multi_update(pts)
        m2 = B.multi_update(pts[::-1].copy())
    finally:
        B.CANON_SIGNS = old
    assert m1.nells == m2.nells > 5
    for e in m1.ells:
        np.testing.assert_array_equal(e.axes, B.canon_cols(e.axes))
