"""Oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): NumPy
restatement of dynesty's proposal generators.

``ref:`` citations are relative to /root/reference/py/dynesty/.  Every
function takes the NumPy ``Generator`` explicitly and consumes it in exactly
the reference's order, so a same-seed comparison with
the reference -- and with the device PCG64/ziggurat streams -- is meaningful.

Each sampler returns a plain dict: u, v, logl, ncalls + the counters the
reference packs into ``tuning_info`` / ``proposal_stats``.
"""
import numpy as np
from numpy import linalg

from .bounding_ref import randsphere, ell_sample, multi_sample

N_EXPAND_THRESHOLD = 1000  # ref: internal_samplers.py:1096


def unitcheck(u, nonbounded=None):
    """ref: utils.py:1036-1050."""
    if nonbounded is None:
        return u.min() > 0 and u.max() < 1
    unb = u[nonbounded]
    ub = u[~nonbounded]
    return (unb.min() > 0 and unb.max() < 1 and ub.min() > -0.5
            and ub.max() < 1.5)


def apply_reflect(u):
    """ref: utils.py:1053-1078."""
    even = np.mod(u, 2) < 1
    u[even] = np.mod(u[even], 1)
    u[~even] = 1 - np.mod(u[~even], 1)
    return u


def propose_ball(u, scale, axes, ncdim, rng, periodic=None, reflective=None,
                 nonbounded=None):
    """One rwalk proposal; returns (u_prop or None, fail).
    ref: internal_samplers.py:989-1035."""
    n = len(u)
    u_prop = np.zeros(n)
    u_prop[ncdim:] = rng.random(n - ncdim)
    dr = randsphere(ncdim, rng)
    du = np.dot(axes, dr)
    u_prop[:ncdim] = u[:ncdim] + scale * du
    if periodic is not None:
        u_prop[periodic] = np.mod(u_prop[periodic], 1)
    if reflective is not None:
        u_prop[reflective] = apply_reflect(u_prop[reflective])
    if unitcheck(u_prop, nonbounded):
        return u_prop, False
    return None, True


def rwalk(u, loglstar, axes, scale, prior_transform, loglikelihood, rng,
          walks, periodic=None, reflective=None, nonbounded=None):
    """ref: internal_samplers.py:866-986 (generic_random_walk)."""
    ncdim = axes.shape[0]
    n_accept = n_reject = ncall = 0
    v = logl = None
    while ncall < walks:
        u_prop, fail = propose_ball(u, scale, axes, ncdim, rng, periodic,
                                    reflective, nonbounded)
        if fail:
            n_reject += 1
            ncall += 1
            continue
        v_prop = prior_transform(u_prop)
        logl_prop = loglikelihood(v_prop)
        ncall += 1
        if logl_prop > loglstar:
            u, v, logl = u_prop, v_prop, logl_prop
            n_accept += 1
        else:
            n_reject += 1
    if n_accept == 0:
        v = prior_transform(u)
        logl = loglikelihood(v)
    return dict(u=u, v=v, logl=logl, ncalls=ncall, accept=n_accept,
                reject=n_reject, scale=scale)


def _doubling_accept(x1, F, loglstar, L, R, fL, fR):
    """ref: internal_samplers.py:1038-1072 (Neal 2003, algorithm 6)."""
    lhat, rhat = L, R
    f_lhat, f_rhat = fL, fR
    D = False
    while rhat - lhat > 1.1:
        M = (lhat + rhat) / 2.
        if (0 < M <= x1) or (x1 < M <= 0):
            D = True
        if x1 < M:
            rhat = M
            f_rhat = F(rhat)[1]
        else:
            lhat = M
            f_lhat = F(lhat)[1]
        if D and loglstar >= f_lhat and loglstar >= f_rhat:
            return False
    return True


def slice_step(u, direction, nonperiodic, loglstar, loglikelihood,
               prior_transform, doubling, rng):
    """ref: internal_samplers.py:1075-1206 (generic_slice_step).

    Returns (u_prop, v_prop, logl_prop, nc, n_expand, n_contract, warn)."""
    nc = n_expand = n_contract = 0
    n = len(u)
    rand0 = rng.random()
    dirlen = linalg.norm(direction)
    maxlen = np.sqrt(n) / 2.
    dirnorm = dirlen / maxlen if dirlen > maxlen else 1
    direction = direction / dirnorm

    def F(x):
        nonlocal nc
        u_new = u + x * direction
        if unitcheck(u_new, nonperiodic):
            logl = loglikelihood(prior_transform(u_new))
        else:
            logl = -np.inf
        nc += 1
        return u_new, logl

    left = -rand0
    right = 1 - rand0
    f_left = F(left)[1]
    f_right = F(right)[1]
    warn = False
    L = R = fL = fR = None
    if not doubling:
        while f_left > loglstar:
            left -= 1
            f_left = F(left)[1]
            n_expand += 1
        while f_right > loglstar:
            right += 1
            f_right = F(right)[1]
            n_expand += 1
        if n_expand > N_EXPAND_THRESHOLD:
            warn = True
    else:
        K = 1
        while f_left > loglstar or f_right > loglstar:
            if rng.random() < 0.5:
                left -= (right - left)
                f_left = F(left)[1]
            else:
                right += (right - left)
                f_right = F(right)[1]
            n_expand += K
            K *= 2
        L, R, fL, fR = left, right, f_left, f_right
    while True:
        width = right - left
        x = left + rng.random() * width
        u_prop, logl_prop = F(x)
        n_contract += 1
        if logl_prop > loglstar and (not doubling or _doubling_accept(
                x, F, loglstar, L, R, fL, fR)):
            break
        if x < 0:
            left = x
        elif x > 0:
            right = x
        else:
            raise RuntimeError("Slice sampler has failed to find a valid point")
    v_prop = prior_transform(u_prop)
    return u_prop, v_prop, logl_prop, nc, n_expand, n_contract, warn


def rslice(u, loglstar, axes, scale, prior_transform, loglikelihood, rng,
           slices, nonperiodic=None, doubling=False):
    """ref: internal_samplers.py:745-855 (RSliceSampler.sample)."""
    n = len(u)
    nc = n_expand = n_contract = 0
    warn_set = False
    v = logl = None
    for _ in range(slices):
        drhat = rng.standard_normal(size=n)
        drhat /= linalg.norm(drhat)
        direction = np.dot(axes, drhat) * scale
        (u, v, logl, nc1, ne1, nct1, warn) = slice_step(
            u, direction, nonperiodic, loglstar, loglikelihood,
            prior_transform, doubling, rng)
        nc += nc1
        n_expand += ne1
        n_contract += nct1
        if warn and not doubling:
            doubling = True
            warn_set = True
    return dict(u=u, v=v, logl=logl, ncalls=nc, n_expand=n_expand,
                n_contract=n_contract, expansion_warning_set=warn_set)


def pslice(u, loglstar, axes, scale, prior_transform, loglikelihood, rng,
           slices, nonperiodic=None, doubling=False):
    """Principal-axes slice sampling.
    ref: internal_samplers.py:593-709 (SliceSampler.sample)."""
    n = len(u)
    nc = n_expand = n_contract = 0
    warn_set = False
    v = logl = None
    saxes = scale * axes.T
    for _ in range(slices):
        idxs = np.arange(n)
        rng.shuffle(idxs)
        for idx in idxs:
            (u, v, logl, nc1, ne1, nct1, warn) = slice_step(
                u, saxes[idx], nonperiodic, loglstar, loglikelihood,
                prior_transform, doubling, rng)
            nc += nc1
            n_expand += ne1
            n_contract += nct1
            if warn and not doubling:
                warn_set = True
                doubling = True
    return dict(u=u, v=v, logl=logl, ncalls=nc, n_expand=n_expand,
                n_contract=n_contract, expansion_warning_set=warn_set)


def unif_bound(loglstar, bound_draw, prior_transform, loglikelihood, rng,
               ndim, ncdim, nonbounded=None):
    """Uniform sampling inside a bound until logl > loglstar.

    ``bound_draw(rng)`` returns one point of the bound (ncdim,).
    ref: internal_samplers.py:243-340 (UniformBoundSampler.sample)."""
    nc = 0
    ntries = 0
    if nonbounded is not None:
        nonbounded = nonbounded[:ncdim]
    while True:
        u = bound_draw(rng)
        if not unitcheck(u, nonbounded):
            ntries += 1
            continue
        ntries = 0
        if ncdim != ndim:
            u = np.concatenate((u, rng.uniform(size=(ndim - ncdim))))
        v = prior_transform(np.asarray(u))
        logl = loglikelihood(np.asarray(v))
        nc += 1
        if logl > loglstar:
            break
    return dict(u=u, v=v, logl=logl, ncalls=nc, n_proposals=ntries)


def unif_single(ell):
    """bound_draw for one ellipsoid. ref: bounding.py:321-334."""
    return lambda rng: ell_sample(ell, rng)


def unif_multi(mell):
    """bound_draw for a union of ellipsoids. ref: bounding.py:592-606."""
    return lambda rng: multi_sample(mell, rng)[0]


def unitcube(loglstar, prior_transform, loglikelihood, rng, ndim):
    """ref: internal_samplers.py:364-441 (UnitCubeSampler.sample)."""
    nc = 0
    while True:
        u = rng.uniform(size=ndim)
        v = prior_transform(np.asarray(u))
        logl = loglikelihood(np.asarray(v))
        nc += 1
        if logl > loglstar:
            break
    return dict(u=u, v=v, logl=logl, ncalls=nc, n_proposals=nc)
