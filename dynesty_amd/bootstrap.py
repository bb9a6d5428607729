"""Bootstrap expansion factor (reference bounding.py:1593-1648
_bootstrap_points + _ellipsoid_bootstrap_expand), with the geometry on the
device: every replica is one rebuild launch on the resampled points plus one
membership launch for the left-out points."""
import numpy as np

from .backend import get_backend


def _split(points, seed):
    """bounding.py:1593-1616 (index bookkeeping only)."""
    rstate = seed if isinstance(seed, np.random.Generator) else \
        np.random.Generator(np.random.PCG64(seed))
    n = points.shape[0]
    idxs = rstate.integers(n, size=n)
    sel = np.zeros(n, dtype=bool)
    sel[np.unique(idxs)] = True
    n_in = sel.sum()
    if n_in < 2:
        sel[:2] = True
    if n_in > n - 1:
        sel[0] = False
    return points[sel], points[~sel]


def resample_mask(n, seed):
    """The `sel_in` mask of bounding.py:1593-1616 for n points."""
    rstate = seed if isinstance(seed, np.random.Generator) else \
        np.random.Generator(np.random.PCG64(seed))
    idxs = rstate.integers(n, size=n)
    sel = np.zeros(n, dtype=bool)
    sel[np.unique(idxs)] = True
    n_in = sel.sum()
    if n_in < 2:
        sel[:2] = True
    if n_in > n - 1:
        sel[0] = False
    return sel


def resample_masks(n, rstate, bootstrap):
    """(bootstrap, n) masks from get_seed_sequence(rstate, bootstrap)
    (utils.py:1002-1009)."""
    seeds = np.random.SeedSequence(rstate.integers(0, 2**63 - 1,
                                                   size=4)).spawn(bootstrap)
    return np.array([resample_mask(n, s) for s in seeds])


def expand_one(multi, points, seed):
    """bounding.py:1619-1648."""
    be = get_backend()
    pin, pout = _split(points, seed)
    res = be.rebuild(np.ascontiguousarray(pin), multi=multi)
    _, _, quad = be.contains(np.ascontiguousarray(pout), res["ctrs"],
                             res["ams"], mode=0, want_quad=True)
    dists = np.sqrt(quad.min(axis=1))
    return max(1., float(np.max(dists)))


def bootstrap_expand(points, rstate, bootstrap, multi, pool=None):
    """max over `bootstrap` replicas (bounding.py:381-400 / 688-703); seeds as
    utils.get_seed_sequence (utils.py:1002-1009).  All replicas are rebuilt by
    ONE ragged batched launch when the backend offers it."""
    seeds = np.random.SeedSequence(rstate.integers(0, 2**63 - 1,
                                                   size=4)).spawn(bootstrap)
    be = get_backend()
    if not hasattr(be, "rebuild_many"):
        return max(expand_one(multi, points, s) for s in seeds)
    splits = [_split(points, s) for s in seeds]
    results = be.rebuild_many([np.ascontiguousarray(pin) for pin, _ in splits],
                              multi=multi)
    expand = 1.
    for (pin, pout), res in zip(splits, results):
        _, _, quad = be.contains(np.ascontiguousarray(pout), res["ctrs"],
                                 res["ams"], mode=0, want_quad=True)
        expand = max(expand, float(np.sqrt(quad.min(axis=1)).max()))
    return expand
