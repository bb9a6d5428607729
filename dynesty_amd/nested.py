"""A compact static nested-sampling driver over the device hot path.

This is the *caller* of the hot path, written so that end-to-end runs (logZ
checks, ensemble benchmark) are possible where dynesty itself is not installed
(the GPU box).  It follows the semantics of dynesty's static sampler with a
proposal queue (reference sampler.py:932-1212 main loop, :676-778 queue
handling, :625-674 bound-update policy, utils.py:1411-1492 evidence
integration) but is an independent, much smaller implementation: static runs
only, device problems only, no checkpointing, no dynamic batches.  With
dynesty installed, use the drop-in classes of ``dynesty_amd.dropin`` instead.

Semantics kept from the reference because they decide the statistics:
  * K = queue_size proposals are generated against the same loglstar and
    consumed one per iteration; a proposal whose logl no longer beats the
    current loglstar is discarded (sampler.py:741-776);
  * the run starts in the unit cube and switches to the bound once the
    efficiency drops below 10% after 2*nlive calls (dynesty.py first_update
    defaults); the bound is rebuilt every update_interval calls and enlarged by
    1.25 in volume (sampler.py:493-510);
  * rwalk scale tuning: scale *= exp((facc_obs - facc)/ncdim/facc) per queue
    fill (internal_samplers.py:460-493); slice tuning per tune_slice (:1209-1239);
  * ln X decreases by ln((N+1)/N) per iteration; trapezoid weights; the final
    live points are appended (sampler.py:780-930).
"""
import heapq
import math

import numpy as np
from scipy.special import logsumexp

from . import bounding
from .backend import get_backend


class RunResult(dict):
    __getattr__ = dict.get


def _integrate_full(logl, logvol):
    """Per-point ln weights, cumulative ln Z, information and var[ln Z] from
    ordered (logl, logvol) with the trapezoid rule (the arithmetic of
    utils.compute_integrals, utils.py:1411-1467)."""
    logl = np.asarray(logl)
    logvol = np.asarray(logvol)
    lpad = np.concatenate([[-1.e300], logl])
    vpad = np.concatenate([[0.], logvol])
    # ln(X_{i-1} - X_i) and the trapezoid factor 1/2
    logdvol = vpad[:-1] + np.log1p(-np.exp(vpad[1:] - vpad[:-1])) + math.log(.5)
    logwt = np.logaddexp(lpad[1:], lpad[:-1]) + logdvol
    logz = np.logaddexp.accumulate(logwt)
    # incomplete information H_x = int_0^x L/Z ln L dX - Z_x/Z ln Z, normalised by the FINAL
    # Z, and var[ln Z] = |cumsum dH * dlnX| (utils.py:1451-1466)
    lz = logz[-1]
    w0 = np.exp(lpad[:-1] - lz + logdvol)
    w1 = np.exp(lpad[1:] - lz + logdvol)
    with np.errstate(invalid='ignore'):
        part = np.cumsum(np.where(w0 > 0, w0 * lpad[:-1], 0.) +
                         np.where(w1 > 0, w1 * lpad[1:], 0.))
    saved_h = part - lz * np.exp(logz - lz)
    dh = np.diff(saved_h, prepend=0)
    dlogvol = -np.diff(vpad)
    logzvar = np.abs(np.cumsum(dh * dlogvol))
    return logwt, logz, saved_h, logzvar


def _integrate(logl, logvol):
    """ln weights, cumulative ln Z, final information and final var[ln Z]."""
    logwt, logz, h, logzvar = _integrate_full(logl, logvol)
    return logwt, logz, float(h[-1]), float(logzvar[-1])


def _logaddexp(x, y):
    # np.logaddexp's formula on Python floats (a NumPy scalar call per iteration was a third of the
    # host loop's time)
    if x == y:
        return x + 0.6931471805599453
    d = x - y
    if d > 0:
        return x + math.log1p(math.exp(-d))
    return y + math.log1p(math.exp(d))


def run_static(prob, nlive=500, bound='multi', sample='rwalk', queue_size=None,
               walks=None, slices=None, rstate=None, dlogz=0.01, enlarge=None,
               bootstrap=None,
               maxiter=None, first_update_min_ncall=None,
               first_update_min_eff=10., verbose=False):
    """One static nested-sampling run on the device.  Returns a RunResult with
    logz, logzerr, niter, ncall, samples_u, samples_logl, logwt, nbound and the
    reference's per-point bookkeeping (sampler.py:1165-1182, 870-890): samples_id
    (live slot), samples_it (iteration at which the point was proposed, counted
    from 1; 0 = initial point), samples_nc (likelihood calls spent on the point's
    replacement; 1 for the final live points)."""
    be = get_backend()
    if rstate is None:
        rstate = np.random.default_rng()
    # dynesty.py:186-193: uniform sampling bootstraps the bound (5 replicas) instead of
    # enlarging it; everything else enlarges by 1.25 in volume
    if bootstrap is None:
        bootstrap = 5 if sample == 'unif' else 0
    if enlarge is None:
        enlarge = 1.0 if sample == 'unif' else 1.25
    nd = prob.ndim
    K = int(queue_size or max(1, nlive // 4))
    if walks is None:
        walks = nd + 20  # dynesty.py:128
    if slices is None:
        slices = 3 + nd if sample == 'rslice' else 3
    if first_update_min_ncall is None:
        first_update_min_ncall = 2 * nlive
    ratio = dict(unif=1, rwalk=walks, rslice=slices, slice=slices * nd)[sample]
    update_interval = max(1, round(ratio * nlive))
    facc = min(1., max(1. / max(2, walks), 0.5))

    def spawn_states(k):
        # utils.get_seed_sequence: 4 ints below 2**63-1 -> SeedSequence children
        ent = rstate.integers(0, 2**63 - 1, size=4)
        return be.seed_children(ent, 0, k)

    # ---- initial live points: uniform in the cube ----
    live_u = rstate.random((nlive, nd))
    live_v, live_logl = be.problem_eval(prob, live_u)
    ncall = nlive
    it = 0
    bnd = None
    unit_cube = True
    scale = 1.0
    doubling = False
    ncall_last_update = 0
    nbound = 0
    logvol = 0.
    dlv = math.log((nlive + 1.) / nlive)
    dead_u, dead_logl, dead_logvol = [], [], []
    dead_id, dead_it, dead_nc = [], [], []
    live_it = np.zeros(nlive, dtype=np.int64)
    nc_acc = 0  # calls of the entries popped since the last death (sampler.py:1141: 'nc')
    logz = -1.e300
    hist = dict(acc=0, rej=0, nexp=0, ncon=0)

    def rebuild():
        nonlocal bnd, nbound
        if bnd is None:
            bnd = dict(multi=bounding.HipMultiEllipsoid,
                       single=bounding.HipEllipsoid,
                       balls=bounding.HipRadFriends,
                       cubes=bounding.HipSupFriends)[bound](nd)
        bnd.update(live_u, rstate=rstate, bootstrap=bootstrap)
        if enlarge != 1.:
            bnd.scale_to_logvol(bnd.logvol + math.log(enlarge))
        nbound += 1

    def fill(loglstar):
        """One queue fill: K proposals against loglstar (one launch)."""
        nonlocal scale, doubling
        states = spawn_states(K)
        if unit_cube:
            out = be.unif_batch(prob, loglstar, states)
            return out, None
        if sample == 'unif':
            if bound in ('balls', 'cubes'):
                # the shapes sit on the CURRENT live points (sampler.py:483-484)
                out = be.unif_friends_batch(prob, loglstar, states, live_u,
                                            bound, bnd.axes, bnd.axes_inv)
            elif bound == 'multi':
                out = be.unif_batch(prob, loglstar, states, ctrs=bnd.ctrs,
                                    axes=bnd.axes_ells, ams=bnd.ams,
                                    logvol_ells=bnd.logvol_ells)
            else:
                out = be.unif_batch(prob, loglstar, states, ctrs=bnd.ctr,
                                    axes=bnd.axes)
            return out, None
        above = np.nonzero(live_logl > loglstar)[0]
        if len(above) == 0:
            raise RuntimeError('No live points are above loglstar.')
        start = rstate.choice(above, size=K)
        if bound == 'multi':
            probs = np.exp(bnd.logvol_ells - bnd.logvol)
            fidx = np.minimum(np.searchsorted(np.cumsum(probs),
                                              rstate.random(K)),
                              bnd.nells - 1).astype(np.int32)
            frames = bnd.axes_ells
        else:
            fidx, frames = None, bnd.axes[None]
        if sample == 'rwalk':
            out = be.rwalk_batch(prob, live_u[start], frames, scale, loglstar,
                                 walks, states, axes_idx=fidx)
            out["ncalls"] = np.full(K, walks)
            hist["acc"] += int(out["accept"].sum())
            hist["rej"] += int(out["reject"].sum())
            tot = hist["acc"] + hist["rej"]
            scale *= math.exp((hist["acc"] / tot - facc) / nd / facc)
            hist["acc"] = hist["rej"] = 0
        else:
            out = be.slice_batch(prob, live_u[start], frames, scale, loglstar,
                                 slices, states, principal=(sample == 'slice'),
                                 doubling=doubling, axes_idx=fidx)
            hist["nexp"] += int(out["n_expand"].sum())
            hist["ncon"] += int(out["n_contract"].sum())
            if out["expansion_warning_set"].any():
                doubling = True
            ne, nc_ = max(hist["nexp"], 1), hist["ncon"]
            scale *= float(np.clip(ne * 2. / (ne + nc_), 0.5, 2))
            hist["nexp"] = hist["ncon"] = 0
        return out, start

    done = False
    log_nlive = math.log(nlive)
    # min-heap over (logl, slot): the worst live point in O(log N) per iteration
    heap = [(float(l), i) for i, l in enumerate(live_logl)]
    heapq.heapify(heap)
    lmax = float(live_logl.max())
    while not done:
        loglstar = heap[0][0]
        # bound-update policy, evaluated when the queue is empty
        eff = 100. * max(it, 1) / ncall
        if unit_cube:
            if ncall >= first_update_min_ncall and eff < first_update_min_eff:
                unit_cube = False
                rebuild()
                ncall_last_update = ncall
        elif ncall >= ncall_last_update + update_interval:
            rebuild()
            ncall_last_update = ncall
        out, _ = fill(loglstar)
        o_logl = out["logl"].tolist()
        o_nc = [int(x) for x in out["ncalls"]]
        for j in range(K):
            ncall += o_nc[j]
            nc_acc += o_nc[j]
            cur, worst = heap[0]
            if not o_logl[j] > cur:
                continue  # stale proposal: discarded (sampler.py:774-776)
            logvol -= dlv
            dead_u.append(live_u[worst].copy())
            dead_logl.append(cur)
            dead_logvol.append(logvol)
            dead_id.append(worst)
            dead_it.append(live_it[worst])
            dead_nc.append(nc_acc)
            nc_acc = 0
            live_it[worst] = it + 1  # self.it starts at 1 (sampler.py:396, 1182)
            lw = cur + logvol  # coarse running evidence for the stop rule
            logz = _logaddexp(logz, lw - log_nlive)
            live_u[worst] = out["u"][j]
            live_v[worst] = out["v"][j]
            live_logl[worst] = o_logl[j]
            heapq.heapreplace(heap, (o_logl[j], worst))
            if o_logl[j] > lmax:
                lmax = o_logl[j]
            it += 1
            if maxiter is not None and it >= maxiter:
                done = True
                break
            if it % 64 == 0 or j == K - 1:
                dz = _logaddexp(0., lmax + logvol - logz)
                if dz < dlogz:
                    done = True
                    break
        if verbose:
            print(f"it={it} ncall={ncall} logz~{logz:.3f} nbound={nbound} "
                  f"scale={scale:.3f}")
    # ---- add the remaining live points (sampler.py:780-930) ----
    order = np.argsort(live_logl)
    lv_live = logvol + np.log(1. - (np.arange(nlive) + 1.) / (nlive + 1.))
    all_logl = np.concatenate([dead_logl, live_logl[order]])
    all_logvol = np.concatenate([dead_logvol, lv_live])
    all_u = np.concatenate([np.array(dead_u).reshape(-1, nd), live_u[order]])
    logwt, logz_arr, h, logzvar = _integrate(all_logl, all_logvol)
    return RunResult(logz=float(logz_arr[-1]),
                     logzerr=math.sqrt(logzvar), niter=it,
                     ncall=ncall, h=h, nbound=nbound, samples_u=all_u,
                     samples_logl=all_logl, logwt=logwt, scale=scale,
                     eff=100. * it / ncall,
                     samples_id=np.concatenate([np.array(dead_id, dtype=np.int64), order]),
                     samples_it=np.concatenate([np.array(dead_it, dtype=np.int64), live_it[order]]),
                     samples_nc=np.concatenate([np.array(dead_nc, dtype=np.int64),
                                                np.ones(nlive, dtype=np.int64)]))
