"""Synthetic likelihood / prior-transform pairs of the BASELINE configs.

dynesty calls the user's ``prior_transform(u)`` and ``loglikelihood(v)`` once
per proposal, *inside* the proposal loop
(/root/reference/py/dynesty/internal_samplers.py:328-329, 957-958, 1116-1117).
For the device path those two callbacks have to live on the GPU, so the
BASELINE problems are described twice, by the same object:

* ``problem.loglikelihood`` / ``problem.prior_transform`` -- plain NumPy
  callables.  These play the role of the *user's* functions: they are what is
  handed to ``dynesty.NestedSampler`` (initial live points, generic lock-step
  path) and what the test oracle evaluates.
* ``problem.device_spec()`` -- ``(like_id, like_par, prior_id, prior_par)``,
  the description uploaded through ``dh_problem_create`` (include/dynhip.h) and
  evaluated in-kernel by ``dynesty_amd/csrc/problem.h``.

Both sides use the same operation order so that they agree to rounding.

Problem definitions:
  C1  3-D unit Gaussian, prior +-10      demos "Demo 2", tests/test_gau.py
  C2  25-D rho=0.4 correlated Normal     demos/Examples -- 25-D Correlated Normal.ipynb cell 7
  C3  2-D eggbox                         demos/Examples -- Eggbox.ipynb cell 7
  C4  200-D iid Normal, normal prior     demos/Examples -- 200-D Multivariate Normal.ipynb cell 8
"""
import math

import numpy as np
from scipy.special import ndtri

# ids shared with include/dynhip.h
LIKE_GAUSS_IID = 0  # logl = -0.5 * sum(v^2) + c                par = [c]
LIKE_GAUSS_PREC = 1  # logl = -0.5 * v^T P v + c                 par = [c, P row-major]
LIKE_EGGBOX = 2  # logl = (2 + prod cos((2 tmax v - tmax)/2))^5  par = [tmax]

PRIOR_IDENTITY = 0  # v = u
PRIOR_AFFINE = 1  # v = a * (2 u - 1) + b                        par = [a, b]
PRIOR_NORMAL = 2  # v = mu + sigma * ndtri(u)                    par = [mu, sigma]


class Problem:
    """A (loglikelihood, prior_transform) pair known to host *and* device."""

    def __init__(self, ndim, like_id, like_par, prior_id, prior_par,
                 logz_truth=None, name=""):
        self.ndim = int(ndim)
        self.like_id = int(like_id)
        self.like_par = np.ascontiguousarray(like_par, dtype=np.float64)
        self.prior_id = int(prior_id)
        self.prior_par = np.ascontiguousarray(prior_par, dtype=np.float64)
        self.logz_truth = logz_truth
        self.name = name
        if like_id == LIKE_GAUSS_PREC:
            self._prec = self.like_par[1:].reshape(ndim, ndim)

    # -- host callables (the "user functions") ------------------------------
    def prior_transform(self, u):
        u = np.asarray(u, dtype=np.float64)
        if self.prior_id == PRIOR_IDENTITY:
            return u.copy()
        if self.prior_id == PRIOR_AFFINE:
            a, b = self.prior_par
            return a * (2.0 * u - 1.0) + b
        if self.prior_id == PRIOR_NORMAL:
            mu, sigma = self.prior_par
            return mu + sigma * ndtri(u)
        raise ValueError("unknown prior id")

    def loglikelihood(self, v):
        v = np.asarray(v, dtype=np.float64)
        if self.like_id == LIKE_GAUSS_IID:
            return float(-0.5 * np.dot(v, v) + self.like_par[0])
        if self.like_id == LIKE_GAUSS_PREC:
            return float(-0.5 * np.dot(v, self._prec @ v) + self.like_par[0])
        if self.like_id == LIKE_EGGBOX:
            tmax = self.like_par[0]
            t = 2.0 * tmax * v - tmax
            return float((2.0 + np.prod(np.cos(t / 2.0)))**5.0)
        raise ValueError("unknown likelihood id")

    # vectorised versions, (k, ndim) -> (k, ndim) / (k,)
    def prior_transform_many(self, u):
        return self.prior_transform(u)

    def loglikelihood_many(self, v):
        v = np.asarray(v, dtype=np.float64)
        if self.like_id == LIKE_GAUSS_IID:
            return -0.5 * np.einsum('ij,ij->i', v, v) + self.like_par[0]
        if self.like_id == LIKE_GAUSS_PREC:
            return -0.5 * np.einsum('ij,jk,ik->i', v, self._prec,
                                    v) + self.like_par[0]
        if self.like_id == LIKE_EGGBOX:
            tmax = self.like_par[0]
            t = 2.0 * tmax * v - tmax
            return (2.0 + np.prod(np.cos(t / 2.0), axis=1))**5.0
        raise ValueError("unknown likelihood id")

    def device_spec(self):
        return (self.like_id, self.like_par, self.prior_id, self.prior_par)

    def __repr__(self):
        return f"Problem({self.name!r}, ndim={self.ndim})"


def gauss_iid(ndim, prior_halfwidth=10.0, name=None):
    """C1 family: N(0, I) likelihood, uniform prior on [-w, w]^ndim."""
    c = -0.5 * ndim * math.log(2.0 * math.pi)
    truth = -ndim * math.log(2.0 * prior_halfwidth)
    return Problem(ndim, LIKE_GAUSS_IID, [c], PRIOR_AFFINE,
                   [prior_halfwidth, 0.0], logz_truth=truth,
                   name=name or f"gauss_iid{ndim}")


def gauss_corr(ndim=25, rho=0.4, prior_halfwidth=5.0, name=None):
    """C2: unit-variance Normal with uniform off-diagonal correlation ``rho``."""
    cov = np.full((ndim, ndim), rho)
    np.fill_diagonal(cov, 1.0)
    prec = np.linalg.inv(cov)
    prec = 0.5 * (prec + prec.T)
    _, logdet = np.linalg.slogdet(cov)
    c = -0.5 * (ndim * math.log(2.0 * math.pi) + logdet)
    # the prior box truncates a negligible part of the mass (|v|<5 sigma)
    truth = -ndim * math.log(2.0 * prior_halfwidth)
    par = np.concatenate([[c], prec.ravel()])
    return Problem(ndim, LIKE_GAUSS_PREC, par, PRIOR_AFFINE,
                   [prior_halfwidth, 0.0], logz_truth=truth,
                   name=name or f"gauss_corr{ndim}")


def eggbox(ndim=2, tmax=5.0 * math.pi, name=None):
    """C3: eggbox on the unit square (identity prior)."""
    return Problem(ndim, LIKE_EGGBOX, [tmax], PRIOR_IDENTITY, [],
                   logz_truth=235.856 if ndim == 2 else None,
                   name=name or f"eggbox{ndim}")


def gauss_normal_prior(ndim=200, name=None):
    """C4: iid N(0,1) likelihood with an N(0,1) prior via ndtri."""
    c = -0.5 * ndim * math.log(2.0 * math.pi)
    truth = -0.5 * ndim * math.log(2.0 * math.pi) - 0.5 * ndim * math.log(2.0)
    return Problem(ndim, LIKE_GAUSS_IID, [c], PRIOR_NORMAL, [0.0, 1.0],
                   logz_truth=truth, name=name or f"gauss_nprior{ndim}")


BASELINE_PROBLEMS = {
    "C1": lambda: gauss_iid(3, 10.0, "C1"),
    "C2": lambda: gauss_corr(25, 0.4, 5.0, "C2"),
    "C3": lambda: eggbox(2, name="C3"),
    "C4": lambda: gauss_normal_prior(200, "C4"),
}
