"""Oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): NumPy/SciPy
restatement of dynesty's RadFriends / SupFriends bounds (N-balls / N-cubes of
one common shape centred on every live point).

All ``ref:`` citations are relative to /root/reference/py/dynesty/.
A bound is a ``Friends`` record (kind, cov, am, axes, axes_inv, logvol, ctrs)
with the field meaning of the reference's attributes (ref: bounding.py:751-764,
1016-1030): ``axes = sqrtm(cov)`` (symmetric), ``am = pinvh(cov)``,
``axes_inv = pinvh(axes)``; ``kind`` is 'balls' (RadFriends, Euclidean norm) or
'cubes' (SupFriends, max norm).
"""
from dataclasses import dataclass

import numpy as np
from scipy import cluster, spatial
from scipy import linalg as sla

from .bounding_ref import bootstrap_split, randsphere, unit_ball_logvol


@dataclass
class Friends:
    kind: str
    cov: np.ndarray
    am: np.ndarray
    axes: np.ndarray
    axes_inv: np.ndarray
    logvol: float
    ctrs: np.ndarray

    @property
    def ndim(self):
        return self.cov.shape[0]

    def copy(self):
        return Friends(self.kind, self.cov.copy(), self.am.copy(), self.axes.copy(), self.axes_inv.copy(),
                       float(self.logvol), np.array(self.ctrs, copy=True))


def shape_logvol(kind, ndim, am):
    """ln volume of ONE ball / cube.  ref: bounding.py:761-762 (balls:
    logvol_prefactor(n) - 0.5 ln det am), :1027-1028 (cubes: n ln 2 - ...)."""
    sign, detln = np.linalg.slogdet(am)
    if not (sign > 0):
        raise ValueError("singular friends metric")  # ref: _slogdet_checked
    pref = unit_ball_logvol(ndim) if kind == 'balls' else ndim * np.log(2.)
    return pref - 0.5 * detln


def friends_init(kind, ndim, cov=None):
    """ref: bounding.py:751-764, 1016-1030."""
    if cov is None:
        cov = np.identity(ndim)
    am = sla.pinvh(cov)
    axes = sla.sqrtm(cov)
    axes_inv = sla.pinvh(axes)
    return Friends(kind, np.array(cov, dtype=float), am, axes, axes_inv, shape_logvol(kind, ndim, am),
                   np.zeros((0, ndim)))


def cluster_labels(points, am):
    """Single-linkage clusters of `points` cut at Mahalanobis distance 1 in the
    metric `am` (ref: bounding.py:963-976).  Returns fcluster ids (1-based)."""
    distances = spatial.distance.pdist(points, metric='mahalanobis', VI=am)
    linkages = cluster.hierarchy.single(distances)
    return cluster.hierarchy.fcluster(linkages, 1.0, criterion='distance')


def covariance_from_clusters(points, am):
    """Covariance of the points after re-centring every cluster on its own mean
    (ref: bounding.py:960-993); plain np.cov when there is one cluster."""
    ids = cluster_labels(points, am)
    if np.max(ids) == 1:
        return np.cov(points, rowvar=False), 1
    moved = np.empty_like(points)
    i = 0
    for idx in np.unique(ids):
        grp = points[ids == idx, :]
        j = i + len(grp)
        moved[i:j, :] = grp - grp.mean(axis=0).reshape((1, -1))
        i = j
    return np.cov(moved, rowvar=False), int(np.max(ids))


def loo_radius(points_t, kind):
    """Leave-one-out nearest-neighbour distance of every point (ref:
    bounding.py:1687-1702): 2-norm for balls, max-norm for cubes."""
    tree = spatial.KDTree(points_t)
    p = 2 if kind == 'balls' else np.inf
    return tree.query(points_t, k=2, eps=0, p=p)[0][:, 1]


def bootstrap_radius(points_t, kind, seed):
    """Largest distance of a left-out point to its nearest resampled point
    (ref: bounding.py:1651-1684)."""
    pin, pout = bootstrap_split(points_t, seed)
    tree = spatial.KDTree(pin)
    p = 2 if kind == 'balls' else np.inf
    return max(tree.query(pout, k=1, eps=0, p=p)[0])


def friends_update(fr, points, seeds=None, use_clustering=True):
    """RadFriends.update / SupFriends.update (ref: bounding.py:876-957,
    1141-1222).  `fr` supplies the metric of the previous bound (`am`) for the
    clustering; `seeds` = list of bootstrap seeds (SeedSequence children or
    ints) or None for the leave-one-out radius.  Returns (new Friends, info)."""
    if use_clustering:
        cov, ncl = covariance_from_clusters(points, fr.am)
    else:
        cov, ncl = np.cov(points, rowvar=False), 1
    am = sla.pinvh(cov)
    axes = sla.sqrtm(cov)
    axes_inv = sla.pinvh(axes)
    points_t = np.dot(points, axes_inv)
    if not seeds:
        radii = loo_radius(points_t, fr.kind)
    else:
        radii = [bootstrap_radius(points_t, fr.kind, s) for s in seeds]
    rmax = max(radii)
    cov = cov * rmax**2
    am = am / rmax**2
    axes = axes * rmax
    axes_inv = axes_inv / rmax
    out = Friends(fr.kind, cov, am, axes, axes_inv, shape_logvol(fr.kind, fr.ndim, am), np.array(points))
    return out, dict(nclusters=ncl, rmax=float(rmax))


def friends_scale_to_logvol(fr, logvol):
    """ref: bounding.py:766-775, 1032-1041."""
    f = np.exp((logvol - fr.logvol) * (1.0 / fr.ndim))
    return Friends(fr.kind, fr.cov * f**2, fr.am / f**2, fr.axes * f, fr.axes_inv / f, float(logvol), fr.ctrs)


def friends_within(fr, x, ctrs=None):
    """Indices of the balls / cubes containing x (ref: bounding.py:777-784,
    1043-1051)."""
    ctrs = fr.ctrs if ctrs is None else ctrs
    t = np.dot(ctrs - x, fr.axes_inv)
    if fr.kind == 'balls':
        return np.where(sla.norm(t, axis=1) <= 1.)[0]
    return np.where(np.max(np.abs(t), axis=1) <= 1.)[0]


def friends_sample(fr, rng, return_q=False):
    """Uniform draw in the UNION (ref: bounding.py:795-831, 1066-1101).
    RNG order per try: balls n normals + 1 uniform, cubes n uniforms; then (more
    than one centre) integers(nctrs); then 1 uniform iff q > 1 and not return_q."""
    n = len(fr.ctrs)
    while True:
        if fr.kind == 'balls':
            ds = randsphere(fr.ndim, rng)
        else:
            ds = rng.uniform(-1, 1, size=fr.ndim)
        dx = np.dot(ds, fr.axes)
        if n == 1:
            x, q = fr.ctrs[0] + dx, 1
        else:
            idx = rng.integers(n)
            x = fr.ctrs[idx] + dx
            q = len(friends_within(fr, x))
        if q == 1 or return_q or rng.random() < (1. / q):
            return (x, q) if return_q else x


def friends_samples(fr, nsamples, rng):
    """ref: bounding.py:833-847."""
    return np.array([friends_sample(fr, rng) for _ in range(nsamples)])
