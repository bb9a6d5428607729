"""The RadFriends / SupFriends oracle (oracle/friends_ref.py) against golden
vectors generated from the REAL reference (tools/make_golden.py friends)."""
import os

import numpy as np
import pytest

import inputs
from oracle import friends_ref as F

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "friends.npz"))
CASES = [(k, n) for k in ("balls", "cubes") for n in inputs.CLOUDS_FRIENDS
         if f"{k}/{n}/u1/cov" in GOLD]


def two_updates(kind, name):
    pts = inputs.cloud(name)
    fr = F.friends_init(kind, pts.shape[1])
    out = []
    for _ in range(2):
        fr, info = F.friends_update(fr, pts)
        out.append(fr)
    return pts, out


@pytest.mark.parametrize("kind,name", CASES)
def test_update_bit_exact(kind, name):
    pts, (f1, f2) = two_updates(kind, name)
    for step, fr in ((1, f1), (2, f2)):
        for k in ("cov", "am", "axes", "axes_inv"):
            np.testing.assert_array_equal(np.real(getattr(fr, k)), GOLD[f"{kind}/{name}/u{step}/{k}"],
                                          err_msg=f"{k} step {step}")
        assert fr.logvol == GOLD[f"{kind}/{name}/u{step}/logvol"]


@pytest.mark.parametrize("kind,name", CASES)
def test_within_samples_scale(kind, name):
    pts, (_, fr) = two_updates(kind, name)
    key = f"{kind}/{name}"
    w = [F.friends_within(fr, x) for x in GOLD[f"{key}/probes"]]
    np.testing.assert_array_equal([len(x) for x in w], GOLD[f"{key}/within_counts"])
    if GOLD[f"{key}/within_idx"].size:
        np.testing.assert_array_equal(np.concatenate(w), GOLD[f"{key}/within_idx"])
    rs = np.random.default_rng(13)
    np.testing.assert_array_equal(F.friends_samples(fr, 12, rs), GOLD[f"{key}/samples"])
    st = rs.bit_generator.state
    after = np.array([st["state"]["state"] >> 64, st["state"]["state"] & (2**64 - 1), st["has_uint32"],
                      st["uinteger"]], dtype=np.uint64)
    np.testing.assert_array_equal(after, GOLD[f"{key}/samples_state_after"])
    rs = np.random.default_rng(14)
    xq = [F.friends_sample(fr, rs, return_q=True) for _ in range(8)]
    np.testing.assert_array_equal(np.array([x for x, q in xq]), GOLD[f"{key}/sample_q_x"])
    np.testing.assert_array_equal([q for x, q in xq], GOLD[f"{key}/sample_q_q"])
    sc = F.friends_scale_to_logvol(fr, fr.logvol + np.log(1.25))
    for k in ("cov", "am", "axes", "axes_inv"):
        np.testing.assert_array_equal(np.real(getattr(sc, k)), GOLD[f"{key}/scaled/{k}"])


@pytest.mark.parametrize("kind,name", CASES)
def test_bootstrap_and_noclustering(kind, name):
    pts = inputs.cloud(name)
    fr = F.friends_init(kind, pts.shape[1])
    fr, _ = F.friends_update(fr, pts)
    # get_seed_sequence(rstate, 3): 4 ints below 2**63-1 -> SeedSequence children (utils.py:1002-1009)
    rstate = np.random.default_rng(9)
    seeds = np.random.SeedSequence(rstate.integers(0, 2**63 - 1, size=4)).spawn(3)
    fb, _ = F.friends_update(fr, pts, seeds=seeds)
    key = f"{kind}/{name}"
    for k in ("cov", "am", "axes", "axes_inv"):
        np.testing.assert_array_equal(np.real(getattr(fb, k)), GOLD[f"{key}/boot/{k}"])
    assert fb.logvol == GOLD[f"{key}/boot/logvol"]
    f0 = F.friends_init(kind, pts.shape[1])
    fn, _ = F.friends_update(f0, pts, use_clustering=False)
    np.testing.assert_array_equal(fn.cov, GOLD[f"{key}/noclust/cov"])
    assert fn.logvol == GOLD[f"{key}/noclust/logvol"]
