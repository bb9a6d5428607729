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

The forced bound update of Sampler.propose_live (sampler.py:484-489: a start point outside the bound rebuilds it
at once) is made per queue fill: the fill's start points go through the batched membership test and one outside
the (enlarged) bound rebuilds it before the frames are drawn (round 4; it matters for few live points in many
dimensions, DESIGN.md section 3.6: the 40-D / 333-live-point case).  The plateau mode's companion stop
(sampler.py:1095-1100, a live set without spread) ends the run with the reference's warning.
"""
import math
import warnings

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


def static_run_logvol(dead_logl, dead_it, live_logl_sorted, live_it_sorted, nlive):
    """ln X of every point of a static run -- dead points in death order, then the final live points
    lowest first -- as the reference's run loop assigns them: ln((N + 1) / N) per iteration
    (sampler.py:1121-1129), except inside a likelihood PLATEAU: when the worst live point shares its
    log-likelihood with others (rwalk hands back its start point if no step was accepted) the next
    `multiplicity` deaths take the constant VOLUME step X / (N + 1) instead (sampler.py:1112-1127,
    1190-1193), and a plateau still open at the end passes its step to the first final points
    (sampler.py:813-830).  Which points were alive at a death follows from the per-point 'it' (iteration at
    which a point was proposed; 0 = initial): alive at death e (0-based) = born at it <= e, not yet dead.
    Returns (dead_logvol, live_logvol)."""
    dead_logl = np.asarray(dead_logl, dtype=np.float64)
    live = np.asarray(live_logl_sorted, dtype=np.float64)
    n, N = len(dead_logl), int(nlive)
    dlv = math.log((N + 1.) / N)
    all_l = np.concatenate([dead_logl, live])
    if len(np.unique(all_l)) == len(all_l):  # no equal values anywhere: no plateau can open
        dead_lv = -dlv * np.arange(1, n + 1)
        return dead_lv, (dead_lv[-1] if n else 0.) + np.log(1. - (np.arange(N) + 1.) / (N + 1.))
    born = np.concatenate([np.asarray(dead_it, dtype=np.int64), np.asarray(live_it_sorted, dtype=np.int64)])
    death = np.concatenate([np.arange(n), np.full(N, n + N)])  # final points: after every death
    vals, counts = np.unique(all_l, return_counts=True)
    dup = set(vals[counts > 1].tolist())
    dead_lv = np.empty(n)
    logvol, pc, plog = 0., 0, 0.
    for e in range(n):
        x = dead_logl[e]
        if pc == 0 and x in dup:
            mult = int(((all_l == x) & (born <= e) & (death >= e)).sum())
            if mult > 1:
                pc, plog = mult, math.log(1. / (N + 1.)) + logvol
        logvol -= -math.log1p(-math.exp(plog - logvol)) if pc else dlv
        dead_lv[e] = logvol
        if pc:
            pc -= 1
    if pc == 0:
        rel = np.log(1. - (np.arange(N) + 1.) / (N + 1.))
    else:
        rel = np.log1p(-((1 + np.arange(pc)) * math.exp(plog - logvol)))
        nrest = N - pc
        rel = np.concatenate([rel, rel[-1] + np.log1p(-(1 + np.arange(nrest)) / (nrest + 1.))])
    return dead_lv, logvol + rel


def _integrate(logl, logvol):
    """ln weights, cumulative ln Z, final information and final var[ln Z]."""
    logwt, logz, h, logzvar = _integrate_full(logl, logvol)
    return logwt, logz, float(h[-1]), float(logzvar[-1])


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
    replacement; 1 for the final live points).

    nlive <= 65535 (slots are 16-bit): the queue is consumed in chunks of <= 512 entries by the device
    operator dh_ns_consume, which keeps the run's keys, their sorted order and the chunk in LDS, or -- above
    nlive = 8192 -- the 513 smallest of them, selected from the keys in global memory (csrc/ns.hip,
    ns_consume_compact)."""
    if nlive < 4 or nlive > 65535:
        raise ValueError(f"run_static: nlive={nlive} is outside the device's queue consumption (4 <= nlive <= 65535)")
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
    nc_acc = 0  # calls of the entries popped since the last death (sampler.py:1141: 'nc')
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

    forced = [False]  # set by fill(): the bound was rebuilt for a start point outside it

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
        # forced bound update (sampler.py:484-489): a start point outside the bound rebuilds it at once
        if bound in ('balls', 'cubes'):
            inside = bnd.overlap_many(live_u[start]) > 0
        else:
            inside = np.asarray(bnd.contains_many(live_u[start]))
        if not inside.all():
            rebuild()
            forced[0] = True
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
            # no step accepted: the start point comes back with its own stored ln L (the reference's re-evaluation
            # gives the same bits, internal_samplers.py:970-975: an exact tie with the live point it copies)
            out["logl"] = np.where(out["accept"] == 0, live_logl[start], out["logl"])
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
    # The queue is consumed by the device operator dh_ns_consume (one workgroup: the loop of sampler.py:
    # 1070-1195 over a whole fill with _new_point's queue rule, the dlogz criterion tested at every
    # iteration); the host keeps the coordinates and applies the surviving replacements in bulk.
    live_l2 = np.ascontiguousarray(live_logl, dtype=np.float64)[None, :]
    live_logl = live_l2[0]
    live_it2 = np.zeros((1, nlive), dtype=np.int32)
    live_it = live_it2[0]
    # logvol, logz, h, logzvar, loglstar of the last dead point, it, ncall, [out] current worst logl
    state = np.array([[0., -1.e300, 0., 0., -1.e300, 0., float(nlive), 0.]])
    plateau = np.zeros((1, 2))  # the reference's plateau mode, carried from chunk to chunk
    loglstar = float(live_logl.min())
    while not done:
        it, ncall = int(state[0, 5]), int(state[0, 6])
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
        # sampler.py:1095-1100; tested here once per fill, before it (the reference tests every iteration: a live set
        # that collapses onto one value in mid-queue is noticed at the next fill; dh_ns_consume's own plateau steps
        # keep the volumes right in between)
        if np.ptp(live_logl) == 0:
            warnings.warn('We have reached the plateau in the likelihood we are stopping sampling')
            break
        out, _ = fill(loglstar)
        if forced[0]:
            forced[0] = False
            ncall_last_update = ncall
        q_logl = np.ascontiguousarray(out["logl"], dtype=np.float64)
        q_nc = np.ascontiguousarray(out["ncalls"], dtype=np.int32)
        pos = 0
        while pos < K and not done:
            # maxiter as the reference and the resident loop count it: the loop stops once its counter EXCEEDS maxiter,
            # i.e. after maxiter + 1 deaths (sampler.py:1076-1083; DH_NS_OPT_MAXITER).  A chunk can produce at most
            # as many deaths as entries popped; chunks of <= 512 entries keep the operator's LDS footprint
            # independent of K
            n = min(K - pos, 512)
            if maxiter is not None:
                n = min(n, maxiter + 1 - int(state[0, 5]))
            res = be.ns_consume(live_l2, q_logl[None, pos:pos + n], q_nc[None, pos:pos + n], state, dlogz,
                                live_it=live_it2, plateau=plateau)
            slots, srcs = res["dead_slot"][0].astype(np.int64), res["dead_src"][0].astype(np.int64) + pos
            ndead = len(slots)
            if ndead:
                # what lived in the slot when it died: the replacement of the previous death of the same
                # slot within this chunk, else the live point from before
                order = np.argsort(slots, kind="stable")
                same = slots[order][1:] == slots[order][:-1]
                prev = np.full(ndead, -1, dtype=np.int64)
                prev[order[1:][same]] = srcs[order[:-1][same]]
                du = np.where((prev < 0)[:, None], live_u[slots], out["u"][np.maximum(prev, 0)])
                dead_u.append(du)
                dead_logl.append(res["dead_logl"][0])
                dead_id.append(slots)
                dead_it.append(res["dead_it"][0].astype(np.int64))
                nc = res["dead_nc"][0].astype(np.int64)
                nc[0] += nc_acc
                dead_nc.append(nc)
                # surviving replacement of every slot = the one of its LAST death
                last = np.ones(ndead, dtype=bool)
                last[order[:-1][same]] = False
                live_u[slots[last]] = out["u"][srcs[last]]
                live_v[slots[last]] = out["v"][srcs[last]]
                nc_acc = int(q_nc[srcs[-1] + 1:pos + n].sum())  # entries popped after the last death
            else:
                nc_acc += int(q_nc[pos:pos + n].sum())
            pos += n
            if res["stopped"][0] or (maxiter is not None and int(state[0, 5]) > maxiter):
                done = True
        loglstar = float(state[0, 7])
        if verbose:
            print(f"it={int(state[0, 5])} ncall={int(state[0, 6])} logz~{state[0, 1]:.3f} nbound={nbound} "
                  f"scale={scale:.3f}")
    it, ncall = int(state[0, 5]), int(state[0, 6])
    dead_u = np.concatenate(dead_u) if dead_u else np.zeros((0, nd))
    dead_logl = np.concatenate(dead_logl) if dead_logl else np.zeros(0)
    dead_id, dead_it, dead_nc = (np.concatenate(x) if x else np.zeros(0, dtype=np.int64)
                                 for x in (dead_id, dead_it, dead_nc))
    # ---- add the remaining live points (sampler.py:780-930) ----
    order = np.argsort(live_logl)
    dead_logvol, lv_live = static_run_logvol(dead_logl, dead_it, live_logl[order], live_it[order], nlive)
    all_logl = np.concatenate([dead_logl, live_logl[order]])
    all_logvol = np.concatenate([dead_logvol, lv_live])
    all_u = np.concatenate([dead_u.reshape(-1, nd), live_u[order]])
    logwt, logz_arr, h, logzvar = _integrate(all_logl, all_logvol)
    return RunResult(logz=float(logz_arr[-1]),
                     logzerr=math.sqrt(logzvar), niter=it,
                     ncall=ncall, h=h, nbound=nbound, samples_u=all_u,
                     samples_logl=all_logl, logwt=logwt, scale=scale,
                     eff=100. * it / ncall,
                     samples_id=np.concatenate([dead_id, order]),
                     samples_it=np.concatenate([dead_it, live_it[order].astype(np.int64)]),
                     samples_nc=np.concatenate([dead_nc, np.ones(nlive, dtype=np.int64)]))
