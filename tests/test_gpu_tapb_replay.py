"""Tap B on the hardware without dynesty on the box: the backend calls that the REAL dynesty NestedSampler
made through the drop-in classes (recorded in the build container by tools/make_trace.py; arguments only,
tests/golden/tapb_*.npz) are replayed against the HIP backend, and every return is held to the oracle
backend's on the same arguments -- counters, indices and generator words exactly, coordinates to 2e-12,
log-likelihoods to 1e-11, ellipsoids to 1e-9, and MultiEllipsoid lists in the SAME ORDER (the oracle backend
runs with the device's eigenvector-sign convention).  Together with tests/test_same_seed_e2e.py (real
dynesty with its own classes == real dynesty with the drop-in on the oracle backend, slot for slot) this
closes the chain reference -> drop-in -> device for whole runs.
"""
import os
import time

import numpy as np
import pytest

import tapb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from dynesty_amd import _lib
    return _lib.Context(0)


@pytest.mark.parametrize("tag", ["c1", "c2s", "egg"])
def test_replay_equals_oracle_backend(ctx, tag):
    from oracle_backend import OracleBackend
    meta, calls = tapb.load_trace(os.path.join(GOLD, f"tapb_{tag}.npz"))
    assert sum(meta["calls"].values()) == len(calls)
    t_dev, t_cpu = {}, {}
    want = tapb.replay(calls, OracleBackend(canon=True), t_cpu)
    got = tapb.replay(calls, ctx, t_dev)
    nmulti = 0
    for (name, args, kwargs), g, w in zip(calls, got, want):
        tapb.compare(name, g, w)
        if name == "rebuild" and isinstance(w, dict) and w["nells"] > 1:
            nmulti += 1
    if tag == "egg":
        assert nmulti >= 3  # many-ellipsoid lists, order included
    # every sampler call of the run went through: the proposals of all queue fills
    n_prop = sum(1 for c in calls if c[0] in ("rwalk_batch", "slice_batch", "unif_batch"))
    assert n_prop == meta["calls"].get("rwalk_batch", 0) + meta["calls"].get("slice_batch", 0) + \
        meta["calls"].get("unif_batch", 0) > 5
    print(f"\n[tap B replay {tag}] {len(calls)} backend calls of a {meta['niter']}-iteration dynesty run: "
          f"device {sum(t_dev.values()):.3f} s, oracle (CPU) {sum(t_cpu.values()):.3f} s; "
          f"by method (device): " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in sorted(t_dev.items())))


def test_replay_full_size_c2_timing(ctx):
    """BASELINE C2 at full size (nlive 2000, queue 512): the first 10 bounded queue fills of the real run,
    replayed for the tap-B backend figure; a sample of walkers of every fill against the oracle."""
    path = os.path.join(GOLD, "tapb_c2.npz")
    if not os.path.exists(path):
        pytest.skip("full-size trace not generated")
    from oracle_backend import OracleBackend
    meta, calls = tapb.load_trace(path)
    fills = [c for c in calls if c[0] == "rwalk_batch"]
    assert len(fills) >= 5
    tapb.replay(calls[:3], ctx)  # warm-up
    t_dev = {}
    t0 = time.perf_counter()
    got = tapb.replay(calls, ctx, t_dev)
    wall = time.perf_counter() - t0
    ob = OracleBackend(canon=True)
    nprop = 0
    for (name, args, kwargs), g in zip(calls, got):
        if name == "rwalk_batch":
            prob, u0, axes, scale, loglstar, walks, states = args[:7]
            nprop += len(u0) * walks
            sel = np.arange(0, len(u0), 37)  # 14 walkers per fill on the CPU
            kw = dict(kwargs)
            if kw.get("axes_idx") is not None:
                kw["axes_idx"] = np.asarray(kw["axes_idx"])[sel]
            w = ob.rwalk_batch(prob, u0[sel], axes, scale, loglstar, walks, states[sel], **kw)
            tapb.compare(name, {k: v[sel] for k, v in g.items()}, w)
        elif name in ("rebuild", "scale_to_logvol"):
            tapb.compare(name, g, tapb.replay([(name, args, kwargs)], ob)[0])
    print(f"\n[tap B, C2 full size] {len(calls)} backend calls ({len(fills)} fills of {len(fills[0][1][1])} walkers, "
          f"{nprop} proposals): device-side wall {wall:.3f} s = {nprop / wall / 1e6:.1f} M proposals/s through the "
          f"host-pointer API; by method: " + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in sorted(t_dev.items())))
