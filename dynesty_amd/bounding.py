"""Host-side mirrors of dynesty's ``Ellipsoid`` / ``MultiEllipsoid`` bounds
(reference: /root/reference/py/dynesty/bounding.py:182-731) whose numerics run
on the GPU through libdynhip.

Same attribute names and method signatures as the reference classes, same
exception types.  The canonical state is plain NumPy (so instances pickle and
deep-copy exactly like the reference's: sampler.py:510, utils.py:2321-2355);
no device handle is ever stored on an instance.

Differences that are visible to a caller (all documented in DESIGN.md):
  * ``axes`` columns are ordered by ascending axis length with a fixed sign
    (largest component positive); LAPACK's order is the same, its signs are
    arbitrary.
  * the order of the ellipsoids of a ``HipMultiEllipsoid`` may be a permutation
    of the reference's (the reference's order depends on LAPACK's signs).
  * ``scale_to_logvol`` reuses the stored principal axes instead of calling
    ``eigh`` again.
"""
import math
from collections.abc import Iterable

import bisect

import numpy as np
from scipy.special import logsumexp

from . import _lib
from .backend import get_backend


def logvol_prefactor(n, p=2.):
    """ln volume constant of the unit L^p ball (bounding.py:1271-1285)."""
    p *= 1.
    return (n * math.log(2.) + n * math.lgamma(1. / p + 1.) -
            math.lgamma(n / p + 1))


def improve_covar_mat(covar0, ntries=100, max_condition_number=1e12):
    """bounding.py:1311-1384 on the device (dh_improve_covar_mat): returns
    (good_mat, covar, am, axes).  The loop constants are compiled into the
    kernels, so only the reference's defaults are accepted."""
    if ntries != 100 or max_condition_number != 1e12:
        raise ValueError("improve_covar_mat: the device routine is built for the reference's "
                         "defaults (ntries=100, max_condition_number=1e12)")
    good, cov, am, axes = get_backend().improve_covar_mat(np.asarray(covar0, dtype=np.float64))
    return bool(good[0]), cov[0], am[0], axes[0]


def _draw(rstate, nsamp, ctrs, axes, ams=None, logvol_ells=None,
          return_q=False):
    """Bound.samples on the device from the caller's numpy Generator; the
    generator is left exactly where numpy would have left it."""
    bitgen = rstate.bit_generator
    st = _lib.pcg_state_words(bitgen)
    xs, idxs, qs, out = get_backend().bound_draw(st, nsamp, ctrs, axes, ams,
                                                 logvol_ells, return_q)
    _lib.set_pcg_state_words(bitgen, out)
    return xs, idxs, qs


class HipBound:
    """Common part of the interface (bounding.py:76-122)."""

    def __init__(self, ndim):
        self.logvol = 0
        self.need_centers = False
        self.ndim = ndim


class HipEllipsoid(HipBound):
    """(x - ctr)^T am (x - ctr) = 1.  bounding.py:182-417."""

    def __init__(self, ndim, ctr=None, cov=None, am=None, axes=None,
                 axlens=None, logvol=None):
        super().__init__(ndim)
        if ctr is None:
            # default: bounding.py:203-205; closed form, no eigen-solve needed
            self.ctr = 0.5 * np.zeros(ndim)
            self.cov = np.identity(ndim) * ndim / 4
            a = math.sqrt(ndim / 4)
            self.axlens = np.full(ndim, a)
            self.axes = np.identity(ndim) * a
            self.am = np.identity(ndim) * (4. / ndim)
            self.logvol = logvol_prefactor(ndim) + 0.5 * ndim * math.log(
                ndim / 4)
        elif axlens is not None and logvol is not None:
            # fully specified (as produced by the rebuild kernel)
            self.ctr = np.asarray(ctr)
            self.cov = np.asarray(cov)
            self.am = np.asarray(am)
            self.axes = np.asarray(axes)
            self.axlens = np.asarray(axlens)
            self.logvol = float(logvol)
        else:
            self.ctr = np.asarray(ctr)
            self.cov = np.asarray(cov)
            ax, al, pm, lv = get_backend().ell_from_cov(self.cov)
            self.axlens = al[0]
            self.logvol = float(lv[0])
            self.axes = ax[0] if axes is None else axes
            self.am = pm[0] if am is None else am
        self.funit = 1

    def scale_to_logvol(self, logvol):
        """bounding.py:242-276."""
        cov = np.ascontiguousarray(self.cov, dtype=np.float64)[None].copy()
        am = np.ascontiguousarray(self.am, dtype=np.float64)[None].copy()
        axes = np.ascontiguousarray(self.axes, dtype=np.float64)[None].copy()
        axlens = np.ascontiguousarray(self.axlens, dtype=np.float64)[None].copy()
        lv = np.array([self.logvol], dtype=np.float64)
        get_backend().scale_to_logvol(cov, am, axes, axlens, lv, [logvol])
        self.cov, self.am, self.axes, self.axlens = cov[0], am[0], axes[0], \
            axlens[0]
        self.logvol = logvol

    def major_axis_endpoints(self):
        """bounding.py:278-284."""
        i = np.argmax(self.axlens)
        v = self.axes[:, i]
        return self.ctr - v, self.ctr + v

    def distance(self, x):
        """bounding.py:286-293."""
        _, _, quad = get_backend().contains(np.asarray(x), self.ctr[None],
                                            self.am[None], mode=1,
                                            want_quad=True)
        return np.sqrt(quad[0, 0])

    def distance_many(self, x):
        """bounding.py:295-300."""
        _, _, quad = get_backend().contains(np.asarray(x), self.ctr[None],
                                            self.am[None], mode=1,
                                            want_quad=True)
        return np.sqrt(quad[:, 0])

    def contains(self, x):
        """bounding.py:302-305 (sqrt(q) <= 1) for ONE point: a scalar query of the sampler's serial
        loop (Sampler.propose_live asks it once per queue entry, sampler.py:484-488), answered on the
        host from the instance's NumPy state with the reference's own expression -- a device round trip
        per point (~50 us) was two thirds of the backend time of a replayed dynesty run.  Batches of
        points go to the membership kernel (`contains_many`, `distance_many`)."""
        d = np.asarray(x, dtype=np.float64) - self.ctr
        return bool(np.sqrt(np.einsum('j,jk,k', d, self.am, d)) <= 1.0)

    def contains_many(self, x):
        """Membership of a batch of points (device, dh_contains)."""
        count, _, _ = get_backend().contains(np.asarray(x), self.ctr[None],
                                             self.am[None], mode=1)
        return count > 0

    def sample(self, rstate=None):
        """bounding.py:307-319."""
        return self.samples(1, rstate=rstate)[0]

    def samples(self, nsamples, rstate=None):
        """bounding.py:321-334."""
        xs, _, _ = _draw(rstate, nsamples, self.ctr[None], self.axes[None])
        return xs

    def unitcube_overlap(self, ndraws=10000, rstate=None):
        """bounding.py:336-343."""
        xs = self.samples(ndraws, rstate=rstate)
        nin = np.sum((xs.min(axis=1) > 0) & (xs.max(axis=1) < 1))
        return 1. * nin / ndraws

    def update(self, points, rstate=None, bootstrap=0, pool=None,
               mc_integrate=False):
        """bounding.py:345-414."""
        points = np.asarray(points)
        res = get_backend().rebuild(points, multi=False)
        self.ndim = points.shape[1]
        self.ctr = res["ctrs"][0]
        self.cov = res["covs"][0]
        self.am = res["ams"][0]
        self.axes = res["axes"][0]
        self.axlens = res["axlens"][0]
        self.logvol = float(res["logvol_ells"][0])
        if bootstrap > 0:
            from .bootstrap import bootstrap_expand
            expand = bootstrap_expand(points, rstate, bootstrap, multi=False,
                                      pool=pool)
            if expand > 1.:
                self.scale_to_logvol(self.logvol + self.ndim * np.log(expand))
        if mc_integrate:
            self.funit = self.unitcube_overlap(rstate=rstate)

    def get_random_axes(self, rstate):
        """bounding.py:416-417."""
        return self.axes


class HipMultiEllipsoid(HipBound):
    """Union of ellipsoids.  bounding.py:420-731."""

    def __init__(self, ndim, ells=None, ctrs=None, covs=None):
        if ells is None and ctrs is None:
            ells = [HipEllipsoid(ndim)]
        if ells is not None:
            if (ctrs is None) and (covs is None):
                self._set_from_ells(ells)
            else:
                raise ValueError("You cannot specific both `ells` and "
                                 "(`ctrs`, `covs`)!")
        else:
            if covs is None:
                raise ValueError("You must specify either `ells` or "
                                 "(`ctrs`, `covs`).")
            ctrs = np.asarray(ctrs, dtype=np.float64)
            covs = np.asarray(covs, dtype=np.float64)
            ax, al, pm, lv = get_backend().ell_from_cov(covs)
            self._set_arrays(ctrs, covs, pm, ax, al, lv)
        super().__init__(ndim)
        self.logvol = logsumexp(self.logvol_ells)
        self.funit = 1

    # -- state ----------------------------------------------------------------
    def _set_arrays(self, ctrs, covs, ams, axes, axlens, logvol_ells):
        self.nells = len(ctrs)
        self.ctrs = np.ascontiguousarray(ctrs, dtype=np.float64)
        self.covs = np.ascontiguousarray(covs, dtype=np.float64)
        self.ams = np.ascontiguousarray(ams, dtype=np.float64)
        self.axes_ells = np.ascontiguousarray(axes, dtype=np.float64)
        self.axlens_ells = np.ascontiguousarray(axlens, dtype=np.float64)
        self.logvol_ells = np.ascontiguousarray(logvol_ells, dtype=np.float64)

    def _set_from_ells(self, ells):
        self._set_arrays(np.array([e.ctr for e in ells]),
                         np.array([e.cov for e in ells]),
                         np.array([e.am for e in ells]),
                         np.array([e.axes for e in ells]),
                         np.array([e.axlens for e in ells]),
                         np.array([e.logvol for e in ells]))

    @property
    def ells(self):
        """List of HipEllipsoid views (reference attribute `ells`; used by
        plotting.boundplot via `bound.ells[i]`)."""
        return [
            HipEllipsoid(self.ctrs.shape[1], ctr=self.ctrs[i],
                         cov=self.covs[i], am=self.ams[i],
                         axes=self.axes_ells[i], axlens=self.axlens_ells[i],
                         logvol=self.logvol_ells[i])
            for i in range(self.nells)
        ]

    # -- geometry ---------------------------------------------------------------
    def scale_to_logvol(self, logvol):
        """bounding.py:478-495."""
        if isinstance(logvol, Iterable):
            targets = np.asarray(logvol, dtype=np.float64)
        else:
            targets = self.logvol_ells + (logvol - self.logvol)
        get_backend().scale_to_logvol(self.covs, self.ams, self.axes_ells,
                                      self.axlens_ells, self.logvol_ells,
                                      targets)
        self.logvol = logsumexp(self.logvol_ells)

    def major_axis_endpoints(self):
        """bounding.py:497-500."""
        return np.array([e.major_axis_endpoints() for e in self.ells])

    def _mask(self, x):
        x = np.asarray(x, dtype=np.float64)
        count, mask, _ = get_backend().contains(x, self.ctrs, self.ams, mode=0,
                                                want_mask=True)
        return count, mask

    def within(self, x, j=None):
        """bounding.py:502-511."""
        _, mask = self._mask(x)
        inside = (mask[:, 0] & np.uint64(1)).astype(bool)
        if j is not None:
            inside[j] = False
        return np.nonzero(inside)[0]

    def overlap(self, x, j=None):
        """bounding.py:513-518."""
        return len(self.within(x, j=j))

    def contains(self, x):
        """bounding.py:520-523 (strict <) for ONE point: host scalar query, see HipEllipsoid.contains."""
        delt = np.asarray(x, dtype=np.float64)[None, :] - self.ctrs
        return bool(np.any(np.einsum('ai,aij,aj->a', delt, self.ams, delt) < 1.))

    def contains_many(self, x):
        """Membership of a batch of points in the union (device, dh_contains)."""
        count, _, _ = get_backend().contains(np.asarray(x, dtype=np.float64),
                                             self.ctrs, self.ams, mode=0)
        return count > 0

    # -- draws --------------------------------------------------------------------
    def sample(self, rstate=None, return_q=False):
        """bounding.py:525-590."""
        xs, idxs, qs = _draw(rstate, 1, self.ctrs, self.axes_ells, self.ams,
                             self.logvol_ells, return_q=return_q)
        if return_q:
            return xs[0], int(idxs[0]), int(qs[0])
        return xs[0], int(idxs[0])

    def samples(self, nsamples, rstate=None):
        """bounding.py:592-606."""
        xs, _, _ = _draw(rstate, nsamples, self.ctrs, self.axes_ells, self.ams,
                         self.logvol_ells)
        return xs

    def monte_carlo_logvol(self, ndraws=10000, rstate=None,
                           return_overlap=True):
        """bounding.py:608-630."""
        xs, _, qs = _draw(rstate, ndraws, self.ctrs, self.axes_ells, self.ams,
                          self.logvol_ells, return_q=True)
        invq = 1. / qs
        qsum = invq.sum()
        logvol = np.log(qsum / ndraws) + self.logvol
        if return_overlap:
            inside = (xs.min(axis=1) > 0) & (xs.max(axis=1) < 1)
            return logvol, (invq * inside).sum() / qsum
        return logvol

    # -- rebuild ----------------------------------------------------------------
    def update(self, points, rstate=None, bootstrap=0, pool=None,
               mc_integrate=False):
        """bounding.py:632-724."""
        points = np.asarray(points)
        npoints, ndim = points.shape
        if npoints == 1:
            raise RuntimeError('Cannot compute the bounding ellipsoid of '
                               'a single point.')
        res = get_backend().rebuild(points, multi=True)
        self._set_arrays(res["ctrs"], res["covs"], res["ams"], res["axes"],
                         res["axlens"], res["logvol_ells"])
        self.logvol = logsumexp(self.logvol_ells)
        if bootstrap > 0:
            from .bootstrap import bootstrap_expand
            expand = bootstrap_expand(points, rstate, bootstrap, multi=True,
                                      pool=pool)
            if expand > 1.:
                self.scale_to_logvol(self.logvol_ells + ndim * np.log(expand))
        if mc_integrate:
            self.logvol, self.funit = self.monte_carlo_logvol(
                rstate=rstate, return_overlap=True)

    def get_random_axes(self, rstate):
        """bounding.py:726-731 (one uniform draw, volume-weighted choice).

        dynesty calls this once per queue entry (sampler.py:689-693): the cumulative
        volumes are kept between calls -- same expression, same bits -- for as long as
        the arrays they were formed from are unchanged (they are compared by value: a
        handful of doubles), and an ellipsoid's frame is always the same view object,
        so that a queue's frames deduplicate by identity (samplers._frames)."""
        key = (id(self.axes_ells), self.logvol_ells.tobytes(), float(self.logvol))
        cache = self.__dict__.get('_axes_cache')
        if cache is None or cache[0] != key:
            probs = np.exp(self.logvol_ells - self.logvol)
            cache = (key, np.cumsum(probs).tolist(), list(self.axes_ells))
            self.__dict__['_axes_cache'] = cache
        cum = cache[1]
        # np.searchsorted(..., side='left') on the same doubles
        idx = min(bisect.bisect_left(cum, rstate.random()), len(cum) - 1)
        return cache[2][idx]

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_axes_cache', None)  # views of this instance's arrays, keyed on their id
        return state


# ---------------------------------------------------------------------------
# RadFriends / SupFriends (bounding.py:734-1263)
# ---------------------------------------------------------------------------
def _sym_funcs(cov):
    """(pinvh(cov), sqrtm(cov), pinvh(sqrtm(cov))) of a symmetric PSD matrix from
    one eigendecomposition (what the reference gets from three LAPACK calls,
    bounding.py:755-758)."""
    lam, vec = np.linalg.eigh(np.asarray(cov, dtype=np.float64))
    d = len(lam)
    eps = np.finfo(float).eps
    top = np.abs(lam).max()
    rt = np.sqrt(np.clip(lam, 0., None))
    inv = np.where(np.abs(lam) > d * eps * top, 1. / np.where(lam == 0, 1., lam), 0.)
    rinv = np.where(rt > d * eps * np.sqrt(top), 1. / np.where(rt == 0, 1., rt), 0.)
    return (vec * inv) @ vec.T, (vec * rt) @ vec.T, (vec * rinv) @ vec.T


class _HipFriends(HipBound):
    """A collection of N-balls / N-cubes of one common shape centred on every
    live point; the centres must be set in ``.ctrs`` (the sampler does it,
    sampler.py:483-484)."""
    kind = None

    def __init__(self, ndim, cov=None):
        super().__init__(ndim)
        self.need_centers = True
        if cov is None:
            cov = np.identity(self.ndim)
        self.cov = np.array(cov, dtype=np.float64)
        self.am, self.axes, self.axes_inv = _sym_funcs(self.cov)
        self.logvol = self._shape_logvol()
        self.funit = 1
        self.ctrs = []  # placeholder

    def _shape_logvol(self):
        sign, detln = np.linalg.slogdet(self.am)
        if not sign > 0:
            raise ValueError('Singular matrix')  # bounding.py _slogdet_checked
        pref = logvol_prefactor(self.ndim) if self.kind == 'balls' else \
            self.ndim * math.log(2.)
        return pref - 0.5 * detln

    def scale_to_logvol(self, logvol):
        """bounding.py:766-775, 1032-1041."""
        f = np.exp((logvol - self.logvol) * (1.0 / self.ndim))
        self.cov = self.cov * f**2
        self.am = self.am / f**2
        self.axes = self.axes * f
        self.axes_inv = self.axes_inv / f
        self.logvol = logvol

    # -- membership ---------------------------------------------------------------
    def _whitened(self, x):
        # one point against every shape: (ctrs - x) axes_inv on the host -- a device call here would
        # re-upload all centres for a single point, once per queue entry (sampler.py:485)
        return np.dot(np.asarray(self.ctrs, dtype=np.float64) - np.asarray(x, dtype=np.float64), self.axes_inv)

    def within(self, x):
        """Indices of the shapes containing x (bounding.py:777-784, 1043-1051)."""
        w = self._whitened(x)
        if self.kind == 'balls':
            return np.where(np.linalg.norm(w, axis=1) <= 1.)[0]
        return np.where(np.max(np.abs(w), axis=1) <= 1.)[0]

    def overlap(self, x):
        """bounding.py:786-791, 1053-1059."""
        return len(self.within(x))

    def contains(self, x):
        """bounding.py:793-796, 1061-1064."""
        return self.overlap(x) > 0

    def overlap_many(self, x):
        """overlap() of every row of x in one device launch (workgroup-per-point membership)."""
        counts, _ = get_backend().friends_within(
            np.asarray(self.ctrs, dtype=np.float64), self.kind, self.axes_inv,
            np.atleast_2d(np.asarray(x, dtype=np.float64)))
        return np.asarray(counts)

    # -- draws ------------------------------------------------------------------
    def _draw(self, rstate, nsamp, return_q=False):
        bitgen = rstate.bit_generator
        xs, qs, out = get_backend().friends_draw(
            _lib.pcg_state6(bitgen), nsamp, np.asarray(self.ctrs, dtype=np.float64),
            self.kind, self.axes, self.axes_inv, return_q)
        _lib.set_pcg_state6(bitgen, out)
        return xs, qs

    def sample(self, rstate=None, return_q=False):
        """bounding.py:798-831, 1066-1101."""
        xs, qs = self._draw(rstate, 1, return_q)
        return (xs[0], int(qs[0])) if return_q else xs[0]

    def samples(self, nsamples, rstate=None):
        """bounding.py:833-847, 1103-1117."""
        return self._draw(rstate, nsamples)[0]

    def monte_carlo_logvol(self, ndraws=10000, rstate=None,
                           return_overlap=True):
        """bounding.py:849-874, 1119-1139."""
        xs, qs = self._draw(rstate, ndraws, return_q=True)
        invq = 1. / qs
        qsum = invq.sum()
        logvol = np.log(1. / ndraws * qsum * len(self.ctrs)) + self.logvol
        if return_overlap:
            inside = (xs.min(axis=1) > 0) & (xs.max(axis=1) < 1)
            return logvol, (invq * inside).sum() / qsum
        return logvol

    # -- rebuild ----------------------------------------------------------------
    def update(self, points, rstate=None, bootstrap=0, pool=None,
               mc_integrate=False, use_clustering=True):
        """bounding.py:876-957, 1141-1222: covariance from the re-centred
        single-linkage clusters (in the metric of the previous update), its
        square root / pseudo-inverses, radius = largest nearest-neighbour
        distance in the whitened frame (leave-one-out, or from the left-out to the
        resampled points of `bootstrap` replicas) -- one device call."""
        points = np.asarray(points, dtype=np.float64)
        masks = None
        if bootstrap > 0:
            from .bootstrap import resample_masks
            masks = resample_masks(len(points), rstate, bootstrap)
        res = get_backend().friends_update(
            points, self.kind, am_prev=self.am if use_clustering else None,
            in_masks=masks)
        self.ndim = points.shape[1]
        self.cov, self.am = res["cov"], res["am"]
        self.axes, self.axes_inv = res["axes"], res["axes_inv"]
        self.logvol = float(res["logvol"])
        self.ctrs = points
        if mc_integrate:
            self.funit = self.monte_carlo_logvol(return_overlap=True,
                                                 rstate=rstate)[1]

    def get_random_axes(self, rstate):
        """bounding.py:995-996."""
        return self.axes


class HipRadFriends(_HipFriends):
    """N-balls (Euclidean norm).  bounding.py:734-996."""
    kind = 'balls'


class HipSupFriends(_HipFriends):
    """N-cubes (max norm).  bounding.py:999-1263."""
    kind = 'cubes'
