"""Oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): NumPy/SciPy
restatement of dynesty's ellipsoidal bounding.

All ``ref:`` citations are relative to /root/reference/py/dynesty/.
An ellipsoid is an ``Ell`` record (ctr, cov, am, axes, axlens, logvol) with the
same field meaning as the reference's ``Ellipsoid`` attributes
(ref: bounding.py:201-240): ``axes[:, i]`` is the i-th principal axis,
``am`` the precision matrix, ``logvol`` the log-volume.
"""
import math
from dataclasses import dataclass

import numpy as np
from scipy import linalg as sla
from scipy.cluster.vq import kmeans2
from scipy.special import gammaln, logsumexp

ROUND_DELTA = 1e-3  # ref: bounding.py:1420
MAX_COND = 1e12  # ref: bounding.py:1311
EIG_MULT = 10  # ref: bounding.py:1326
NTRIES = 100  # ref: bounding.py:1311

# LAPACK leaves the sign of every eigenvector open; the device fixes it (largest-magnitude component
# positive).  The sign decides which k-means seed is "cluster 0" (ctr - axis vs ctr + axis, ref:
# bounding.py:278-284, 1505-1510) and with it the ORDER of the children, of the ellipsoid list and of
# every draw that indexes it.  With CANON_SIGNS the oracle applies the device's convention at the same
# place (right after eigh), so that whole runs can be compared seed for seed; tests/refshim.py patches
# the real reference's eigh the same way.  Off by default: the golden vectors hold LAPACK's signs.
CANON_SIGNS = False


def canon_cols(vec):
    """Flip every column so that its largest-magnitude component is positive (first one on ties)."""
    out = np.array(vec, dtype=np.float64)
    for k in range(out.shape[1]):
        i = np.argmax(np.abs(out[:, k]))
        if out[i, k] < 0:
            out[:, k] = -out[:, k]
    return out


@dataclass
class Ell:
    ctr: np.ndarray
    cov: np.ndarray
    am: np.ndarray
    axes: np.ndarray
    axlens: np.ndarray
    logvol: float

    @property
    def ndim(self):
        return self.ctr.shape[0]

    def copy(self):
        return Ell(self.ctr.copy(), self.cov.copy(), self.am.copy(),
                   self.axes.copy(), self.axlens.copy(), float(self.logvol))


def unit_ball_logvol(ndim):
    """ln volume of the unit ndim-ball. ref: bounding.py:1271-1285 (p=2)."""
    p = 2.0
    return (ndim * np.log(2.) + ndim * gammaln(1. / p + 1.) -
            gammaln(ndim / p + 1))


def make_ell(ctr, cov, am=None, axes=None):
    """ref: bounding.py:201-240 (Ellipsoid.__init__)."""
    ctr = np.asarray(ctr)
    cov = np.asarray(cov)
    ndim = ctr.shape[0]
    lam, vec = sla.eigh(cov, check_finite=False)
    if CANON_SIGNS:
        vec = canon_cols(vec)
    if not np.all((lam > 0.) & np.isfinite(lam)):
        raise ValueError("singular ellipsoid covariance")
    axlens = np.sqrt(lam)
    logvol = unit_ball_logvol(ndim) + 0.5 * np.log(lam).sum()
    if axes is None:
        axes = vec * axlens
    if am is None:
        am = (vec * (1. / lam)) @ vec.T
    return Ell(ctr, cov, am, axes, axlens, float(logvol))


def default_ell(ndim):
    """ref: bounding.py:203-205 (ctr=0, cov = I*ndim/4)."""
    return make_ell(0.5 * np.zeros(ndim), np.identity(ndim) * ndim / 4)


def regularize_cov(cov_in):
    """ref: bounding.py:1311-1384 (improve_covar_mat).

    Returns (good, cov, am, axes); ``good`` is True iff the input needed no fix.
    """
    ndim = cov_in.shape[0]
    cov = np.array(cov_in)
    coeffmin = 1e-10
    failed = 0
    trial = 0
    lam = vec = axes = None
    for trial in range(NTRIES):
        failed = 0
        try:
            lam, vec = sla.eigh(cov, check_finite=False)
            if CANON_SIGNS:
                vec = canon_cols(vec)
            top = lam.max()
            bot = lam.min()
            if np.isfinite(lam).all():
                if top <= 0:
                    failed = 2
                elif bot < top / MAX_COND:
                    failed = 1
                else:
                    axes = vec * lam**.5
                    break
            else:
                failed = 2
        except sla.LinAlgError:
            failed = 2
        if failed == 1:
            lam_fix = np.maximum(lam, EIG_MULT * top / MAX_COND)
            cov = (vec * lam_fix) @ vec.T
        elif failed == 2:
            coeff = coeffmin * (1. / coeffmin)**(trial * 1. / (NTRIES - 1))
            cov = (1. - coeff) * cov + coeff * np.eye(ndim)
    if failed > 0:
        cov = np.eye(ndim)
        am = cov.copy()
        axes = cov.copy()
    else:
        am = (vec * (1. / lam)) @ vec.T
    return trial == 0, cov, am, axes


def bounding_ellipsoid(points):
    """ref: bounding.py:1387-1461."""
    npoints, ndim = points.shape
    if npoints == 1:
        raise ValueError("single point")
    ctr = np.mean(points, axis=0)
    cov = np.cov(points, rowvar=False)
    delta = points - ctr
    if ndim == 1:
        cov = np.atleast_2d(cov)
    lim = 1. - ROUND_DELTA
    am = axes = None
    for ipass in range(2):
        good, cov, am, axes = regularize_cov(cov)
        fmax = np.einsum('ij,jk,ik->i', delta, am, delta).max()
        if ipass == 0 and fmax > lim:
            mult = fmax / lim
            cov *= mult
            am /= mult
            axes *= np.sqrt(mult)
        if ipass == 1 and fmax >= 1:
            raise RuntimeError("failed to contain all the points")
        if good:
            break
    return make_ell(ctr, cov, am=am, axes=axes)


def major_axis_endpoints(ell):
    """ref: bounding.py:278-284."""
    i = np.argmax(ell.axlens)
    v = ell.axes[:, i]
    return ell.ctr - v, ell.ctr + v


def split_tree(points, ell, scale=None, trace=None, depth=0):
    """Recursive k=2 split. ref: bounding.py:1464-1563 (_bounding_ellipsoids).

    ``trace`` (optional list) receives one dict per visited node that ran
    k-means: depth, npoints, labels -- used to compare split trees.
    """
    npoints, ndim = points.shape
    min_size = 2 * ndim
    if npoints < min_size * 2:
        return [ell]
    p1, p2 = major_axis_endpoints(ell)
    seeds = np.vstack((p1, p2))
    if scale is None:
        scale = points.std(axis=0)[None, :]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, labels = kmeans2(points / scale, k=seeds / scale, iter=10,
                            minit='matrix', check_finite=False)
    if trace is not None:
        trace.append(dict(depth=depth, npoints=npoints, labels=labels.copy()))
    parts = [points[labels == k, :] for k in (0, 1)]
    if min(parts[0].shape[0], parts[1].shape[0]) < min_size:
        return [ell]
    kids = [bounding_ellipsoid(p) for p in parts]
    nparam = (ndim * (ndim + 3)) // 2
    dec = nparam * np.log(npoints) / npoints
    out = (split_tree(parts[0], kids[0], scale=scale, trace=trace,
                      depth=depth + 1) +
           split_tree(parts[1], kids[1], scale=scale, trace=trace,
                      depth=depth + 1))
    if (np.logaddexp(kids[0].logvol, kids[1].logvol) - ell.logvol) < -dec:
        return out
    if ((logsumexp([e.logvol for e in out]) - ell.logvol) <
            -dec * (len(out) - 1)):
        return out
    return [ell]


@dataclass
class MultiEll:
    """Stacked arrays of a union of ellipsoids. ref: bounding.py:440-476."""
    ells: list
    ctrs: np.ndarray
    covs: np.ndarray
    ams: np.ndarray
    logvol_ells: np.ndarray
    logvol: float

    @property
    def nells(self):
        return len(self.ells)


def stack_ells(ells):
    """ref: bounding.py:469-476, 466."""
    ctrs = np.array([e.ctr for e in ells])
    covs = np.array([e.cov for e in ells])
    ams = np.array([e.am for e in ells])
    lvs = np.array([e.logvol for e in ells])
    return MultiEll(list(ells), ctrs, covs, ams, lvs, float(logsumexp(lvs)))


def multi_quadforms(x, ctrs, ams):
    """(x-c_a)^T A_a (x-c_a) for every ellipsoid. ref: bounding.py:506-507."""
    delt = x[None, :] - ctrs
    return np.einsum('ai,aij,aj->a', delt, ams, delt)


def multi_within(x, ctrs, ams, j=None):
    """ref: bounding.py:502-511 (strict <)."""
    mask = multi_quadforms(x, ctrs, ams) < 1
    if j is not None:
        mask[j] = False
    return np.nonzero(mask)[0]


def multi_contains(x, ctrs, ams):
    """ref: bounding.py:520-523."""
    return bool(np.any(multi_quadforms(x, ctrs, ams) < 1))


def ell_distance(ell, x):
    """ref: bounding.py:286-293."""
    d = x - ell.ctr
    return np.sqrt(np.dot(np.dot(d, ell.am), d))


def ell_distance_many(ell, x):
    """ref: bounding.py:295-300."""
    d = x - ell.ctr[None, :]
    return np.sqrt(np.einsum('ij,jk,ik->i', d, ell.am, d))


def ell_contains(ell, x):
    """ref: bounding.py:302-305 (note: <=, unlike the multi version)."""
    return bool(ell_distance(ell, x) <= 1.0)


def multi_update(points, trace=None):
    """ref: bounding.py:632-686 (MultiEllipsoid.update without bootstrap)."""
    npoints, _ = points.shape
    if npoints == 1:
        raise RuntimeError("single point")
    first = bounding_ellipsoid(points)
    ells = split_tree(points, first, trace=trace)
    mell = stack_ells(ells)
    if not all(multi_contains(p, mell.ctrs, mell.ams) for p in points):
        raise RuntimeError('Rejecting invalid MultiEllipsoid region')
    return mell


def scale_ell_to_logvol(ell, logvol):
    """In place. ref: bounding.py:242-276."""
    ndim = ell.ndim
    logf = logvol - ell.logvol
    max_log_axlen = np.log(np.sqrt(ndim) / 2)
    log_axlen = np.log(ell.axlens)
    if log_axlen.max() < max_log_axlen - logf / ndim:
        f = np.exp(logf / ndim)
        ell.cov = ell.cov * f**2
        ell.am = ell.am * (1. / f**2)
        ell.axlens = ell.axlens * f
        ell.axes = ell.axes * f
    else:
        logfax = np.zeros(ndim)
        left = logf
        nleft = ndim
        lam, vec = sla.eigh(ell.cov, check_finite=False)
        if CANON_SIGNS:
            vec = canon_cols(vec)
        for i in np.argsort(lam)[::-1]:
            delta = max(min(max_log_axlen - log_axlen[i], left / nleft), 0)
            logfax[i] = delta
            left -= delta
            nleft -= 1
        fax = np.exp(logfax)
        lam1 = lam * fax**2
        ell.cov = (vec * lam1) @ vec.T
        ell.am = (vec * (1. / lam1)) @ vec.T
        ell.axlens = ell.axlens * fax
        ell.axes = ell.axes * fax
    ell.logvol = float(logvol)


def scale_multi_to_logvol(mell, logvol):
    """Returns a re-stacked MultiEll. ref: bounding.py:478-495."""
    if np.ndim(logvol) > 0:
        targets = logvol
    else:
        targets = mell.logvol_ells + (logvol - mell.logvol)
    for i, e in enumerate(mell.ells):
        scale_ell_to_logvol(e, targets[i])
    return stack_ells(mell.ells)


# ---------------------------------------------------------------------------
# random draws
# ---------------------------------------------------------------------------
def randsphere(n, rng):
    """Uniform draw in the unit n-ball. ref: bounding.py:1288-1297.
    RNG order: n normals, then one uniform."""
    z = rng.standard_normal(size=n)
    return z * (rng.random()**(1. / n) / sla.norm(z, check_finite=False))


def rand_choice(pb, rng):
    """ref: bounding.py:1300-1308."""
    return min(np.searchsorted(np.cumsum(pb), rng.random()), len(pb) - 1)


def ell_sample(ell, rng):
    """ref: bounding.py:307-319."""
    return ell.ctr + np.dot(ell.axes, randsphere(ell.ndim, rng))


def multi_sample(mell, rng, return_q=False):
    """ref: bounding.py:525-590. Raises RuntimeError on q==0 beyond 1+1e-3."""
    if mell.nells == 1:
        x = ell_sample(mell.ells[0], rng)
        return (x, 0, 1) if return_q else (x, 0)
    probs = np.exp(mell.logvol_ells - mell.logvol)
    while True:
        idx = rand_choice(probs, rng)
        x = ell_sample(mell.ells[idx], rng)
        quad = multi_quadforms(x, mell.ctrs, mell.ams)
        q = (quad < 1).sum()
        if q == 0:
            q = (quad <= 1 + 1e-3).sum()
            if q == 0:
                raise RuntimeError('Ellipsoid check failed q=0')
        if return_q:
            return x, idx, q
        if q == 1 or rng.random() < (1. / q):
            return x, idx


def multi_random_axes(mell, rng):
    """ref: bounding.py:726-731."""
    probs = np.exp(mell.logvol_ells - mell.logvol)
    return mell.ells[rand_choice(probs, rng)].axes


# ---------------------------------------------------------------------------
# bootstrap expansion  (ref: bounding.py:1593-1648)
# ---------------------------------------------------------------------------
def bootstrap_split(points, seed):
    """ref: bounding.py:1593-1616."""
    rng = seed if isinstance(seed, np.random.Generator) else \
        np.random.Generator(np.random.PCG64(seed))
    n = points.shape[0]
    idxs = rng.integers(n, size=n)
    sel = np.zeros(n, dtype=bool)
    sel[np.unique(idxs)] = True
    n_in = sel.sum()
    if n_in < 2:
        sel[:2] = True
    if n_in > n - 1:
        sel[0] = False
    return points[sel], points[~sel]


def bootstrap_expand(multi, points, seed):
    """ref: bounding.py:1619-1648."""
    pin, pout = bootstrap_split(points, seed)
    ell = bounding_ellipsoid(pin)
    if not multi:
        dists = ell_distance_many(ell, pout)
    else:
        ells = split_tree(pin, ell)
        dists = np.min(np.array([ell_distance_many(e, pout) for e in ells]),
                       axis=0)
    return max(1., np.max(dists))
