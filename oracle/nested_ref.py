"""Oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py): restatement of the pieces of
dynesty's static run loop that the device-resident loop (csrc/ns.hip) reproduces.

All ``ref:`` citations are relative to /root/reference/py/dynesty/.  Pinned to the real reference
by tests/golden/nsloop.npz (tools/make_golden.py nsloop: a recorded queue fill of a real
``NestedSampler`` run and the per-iteration evidence history of that run),
tests/test_oracle_nsloop_golden.py.
"""
import math

import numpy as np
from scipy.special import logsumexp


def progress_integration(loglstar, loglstar_new, logz, logzvar, logvol, dlogvol, h):
    """ref: utils.py:1470-1492.  Returns logwt, logz, logzvar, h."""
    logdvol = logsumexp(a=[logvol + dlogvol, logvol], b=[0.5, -0.5])
    logwt = np.logaddexp(loglstar_new, loglstar) + logdvol
    logz_new = np.logaddexp(logz, logwt)
    lzterm = (math.exp(loglstar - logz_new + logdvol) * loglstar +
              math.exp(loglstar_new - logz_new + logdvol) * loglstar_new)
    h_new = lzterm + math.exp(logz - logz_new) * (h + logz) - logz_new
    dh = h_new - h
    return logwt, logz_new, logzvar + dh * dlogvol, h_new


class RunState:
    """The evidence accumulators of Sampler.sample (ref: sampler.py:1040-1066)."""

    def __init__(self, nlive):
        self.nlive = int(nlive)
        self.dlv = math.log((nlive + 1.) / nlive)  # ref: sampler.py:141
        self.h = 0.
        self.logz = -1.e300
        self.logzvar = 0.
        self.logvol = 0.
        self.loglstar = -1.e300
        self.it = 0
        self.ncall = 0
        # likelihood-plateau bookkeeping (ref: sampler.py:1112-1127, 1190-1193)
        self.plateau_mode = False
        self.plateau_counter = 0
        self.plateau_logdvol = 0.

    def copy(self):
        c = RunState(self.nlive)
        c.__dict__.update(self.__dict__)
        return c

    def as_tuple(self):
        return (self.logz, self.logzvar, self.h, self.logvol, self.loglstar, self.it, self.ncall)


def consume_queue(live_logl, queue_logl, queue_ncalls, state, dlogz, plateau=True, live_it=None):
    """One queue fill consumed by the iteration loop (no maxiter/maxcall).

    ref: sampler.py:1070-1195 (stopping rule, worst point, plateau bookkeeping, volume step,
    progress_integration, replacement) with _new_point's queue rule (sampler.py:741-776): entries
    are popped in order, the calls of every popped entry are charged, an entry whose logl does not
    beat the current worst point's is discarded ("stale").  The loop ends when the queue is empty --
    the reference would refill it at that point -- or when the dlogz criterion fires.

    plateau=False leaves out the reference's plateau mode (equal log-likelihoods among the live
    points: rwalk hands back its start point when no step was accepted); that is the documented
    simplification of the device-resident loop, whose volume step is always ln((N + 1) / N).

    live_logl: (N,) in slot order, modified in place.  Returns a dict: dead_logl / dead_slot /
    dead_src (queue index that replaced the slot) in death order, stopped (bool), used (number of
    queue entries popped).  With live_it ((N,) ints, modified in place: iteration at which the
    point in each slot was proposed, ref: sampler.py:1107, 1182 -- self.it starts at 1, :396) also
    dead_it ('it' of saved_run) and dead_nc ('nc': the calls of every entry popped for that
    iteration, ref: sampler.py:1141-1142, 1176); nc_carry = calls popped after the last death."""
    live_logl = np.asarray(live_logl)
    K = len(queue_logl)
    s = state
    dead_logl, dead_slot, dead_src = [], [], []
    dead_it, dead_nc = [], []
    nc = 0
    j = 0
    stopped = False
    while True:
        delta_logz = np.logaddexp(0, np.max(live_logl) + s.logvol - s.logz)  # ref: sampler.py:1071
        if dlogz is not None and delta_logz < dlogz:
            stopped = True
            break
        worst = int(np.argmin(live_logl))  # ref: sampler.py:1107 (lowest index among ties)
        loglstar_new = live_logl[worst]
        # ref: _new_point -- pop until an entry beats the worst point
        found = None
        while j < K:
            cand = j
            j += 1
            s.ncall += int(queue_ncalls[cand])
            nc += int(queue_ncalls[cand])
            if queue_logl[cand] > loglstar_new:
                found = cand
                break
        if found is None:
            break  # queue exhausted: the reference refills here (nothing of this iteration is applied yet)
        if plateau and not s.plateau_mode:  # ref: sampler.py:1112-1119
            nplateau = int((live_logl == loglstar_new).sum())
            if nplateau > 1:
                s.plateau_mode = True
                s.plateau_counter = nplateau
                s.plateau_logdvol = np.log(1. / (s.nlive + 1)) + s.logvol
        if not s.plateau_mode:
            cur_dlv = s.dlv
        else:
            cur_dlv = -np.log1p(-np.exp(s.plateau_logdvol - s.logvol))  # ref: sampler.py:1125
        s.logvol -= cur_dlv  # ref: sampler.py:1129
        _, s.logz, s.logzvar, s.h = progress_integration(s.loglstar, loglstar_new, s.logz, s.logzvar,
                                                         s.logvol, cur_dlv, s.h)
        s.loglstar = loglstar_new
        dead_logl.append(float(loglstar_new))
        dead_slot.append(worst)
        dead_src.append(found)
        live_logl[worst] = queue_logl[found]
        dead_nc.append(nc)
        nc = 0
        if live_it is not None:
            dead_it.append(int(live_it[worst]))
            live_it[worst] = s.it + 1  # RunState.it counts deaths from 0; the reference's self.it from 1
        s.it += 1
        if s.plateau_mode:  # ref: sampler.py:1190-1193
            s.plateau_counter -= 1
            if s.plateau_counter == 0:
                s.plateau_mode = False
    return dict(dead_logl=np.array(dead_logl), dead_slot=np.array(dead_slot, dtype=np.int64),
                dead_src=np.array(dead_src, dtype=np.int64), stopped=stopped, used=j,
                dead_it=np.array(dead_it, dtype=np.int64), dead_nc=np.array(dead_nc, dtype=np.int64), nc_carry=nc)


def add_live_points(live_logl, state):
    """ref: sampler.py:780-930: the remaining live points, lowest first, each with the expected
    volume e^logvol (N + 1 - i) / (N + 1) (a plateau still being worked off keeps its own volume
    steps first).  Returns the final (logz, logzvar, h)."""
    s = state.copy()
    n = s.nlive
    if not s.plateau_mode:
        logvols = np.log(1. - (np.arange(n) + 1.) / (n + 1.))
    else:
        logvols = np.log1p(-((1 + np.arange(s.plateau_counter)) * np.exp(s.plateau_logdvol - s.logvol)))
        nrest = n - s.plateau_counter
        logvols = np.concatenate([logvols, logvols[-1] + np.log1p(-(1 + np.arange(nrest)) / (nrest + 1))])
    dlvs = -np.diff(logvols, prepend=0)
    logvols = logvols + s.logvol
    for i, idx in enumerate(np.argsort(live_logl)):
        lnew = live_logl[idx]
        _, s.logz, s.logzvar, s.h = progress_integration(s.loglstar, lnew, s.logz, s.logzvar,
                                                         logvols[i], dlvs[i], s.h)
        s.loglstar = lnew
    return s.logz, s.logzvar, s.h


def integrate_static_run(dead_logl, live_logl, nlive):
    """ln Z, sqrt(var ln Z), H of a finished static run from its dead points (death order) and final
    live points: the recurrence the reference runs while sampling (sampler.py:1153-1156) followed
    by add_live_points."""
    s = RunState(nlive)
    for lnew in np.asarray(dead_logl, dtype=np.float64):
        s.logvol -= s.dlv
        _, s.logz, s.logzvar, s.h = progress_integration(s.loglstar, lnew, s.logz, s.logzvar,
                                                         s.logvol, s.dlv, s.h)
        s.loglstar = lnew
        s.it += 1
    logz, logzvar, h = add_live_points(np.asarray(live_logl, dtype=np.float64), s)
    return logz, math.sqrt(abs(logzvar)), h


def compute_integrals(logl, logvol):
    """ref: utils.py:1411-1467 (reweight=None).  Returns logwt, logz, logzvar, h per point -- the
    values run_nested stores at the very end (sampler.py:1342-1348) and Results reports."""
    logl = np.asarray(logl, dtype=np.float64)
    logvol = np.asarray(logvol, dtype=np.float64)
    loglstar_pad = np.concatenate([[-1.e300], logl])
    dlogvol = np.diff(logvol, prepend=0)
    logdvol = logvol - dlogvol + np.log1p(-np.exp(dlogvol))
    logdvol2 = logdvol + math.log(0.5)
    dlogvol = -np.diff(logvol, prepend=0)
    saved_logwt = np.logaddexp(loglstar_pad[1:], loglstar_pad[:-1]) + logdvol2
    saved_logz = np.logaddexp.accumulate(saved_logwt)
    logzmax = saved_logz[-1]
    h_part1 = np.cumsum(
        (np.exp(loglstar_pad[1:] - logzmax + logdvol2) * loglstar_pad[1:] +
         np.exp(loglstar_pad[:-1] - logzmax + logdvol2) * loglstar_pad[:-1]))
    saved_h = h_part1 - logzmax * np.exp(saved_logz - logzmax)
    dh = np.diff(saved_h, prepend=0)
    saved_logzvar = np.abs(np.cumsum(dh * dlogvol))
    return saved_logwt, saved_logz, saved_logzvar, saved_h


def static_run_logvol(niter, nlive):
    """ln X of every point of a finished static run WITHOUT likelihood plateaus: dead points
    (sampler.py:1129: logvol -= dlv per iteration) then the final live points (sampler.py:816-836)."""
    dlv = math.log((nlive + 1.) / nlive)
    dead = np.zeros(niter)
    lv = 0.
    for i in range(niter):  # the reference subtracts step by step (not -(i+1) * dlv)
        lv -= dlv
        dead[i] = lv
    live = np.log(1. - (np.arange(nlive) + 1.) / (nlive + 1.)) + lv
    return np.concatenate([dead, live])


def final_results(dead_logl, live_logl, nlive):
    """What Results reports for a finished static run: (logz, logzerr, information) from
    compute_integrals over dead points + sorted final live points."""
    logl = np.concatenate([np.asarray(dead_logl, dtype=np.float64), np.sort(np.asarray(live_logl))])
    _, logz, logzvar, h = compute_integrals(logl, static_run_logvol(len(dead_logl), nlive))
    return float(logz[-1]), float(math.sqrt(logzvar[-1])), float(h[-1])


def logvol_from_record(dead_logl, dead_id, live_logl, nlive):
    """ln X of every point of a FINISHED static run, replayed from its record -- dead points in death order with
    the slot ('id') each lived in, and the final live points by slot -- with the reference loop's volume
    bookkeeping: the worst point dies, nplateau = number of live points sharing its log-likelihood opens the
    plateau mode (sampler.py:1112-1119), the step is ln((N + 1) / N) or the plateau's (sampler.py:1121-1129),
    the counter runs down (sampler.py:1190-1193); then the final live points as add_live_points assigns them
    (sampler.py:813-830).  Returns the volumes of dead points followed by the final live points in ascending
    log-likelihood.  (Independent of dynesty_amd.nested.static_run_logvol, which works from the 'it' column.)"""
    dead_logl = np.asarray(dead_logl, dtype=np.float64)
    dead_id = np.asarray(dead_id, dtype=np.int64)
    final = np.asarray(live_logl, dtype=np.float64)
    n, N = len(dead_logl), int(nlive)
    # what enters a slot when its occupant dies: the next point to die there, or the final occupant
    repl = np.empty(n)
    nxt = final.copy()
    for e in range(n - 1, -1, -1):
        repl[e] = nxt[dead_id[e]]
        nxt[dead_id[e]] = dead_logl[e]
    cur = nxt  # the live set before the first death, by slot
    s = RunState(N)
    out = np.empty(n)
    for e in range(n):
        worst = int(np.argmin(cur))
        assert cur[worst] == dead_logl[e] and cur[dead_id[e]] == dead_logl[e]
        if not s.plateau_mode:
            nplateau = int((cur == cur[worst]).sum())
            if nplateau > 1:
                s.plateau_mode, s.plateau_counter = True, nplateau
                s.plateau_logdvol = np.log(1. / (N + 1)) + s.logvol
        cur_dlv = s.dlv if not s.plateau_mode else -np.log1p(-np.exp(s.plateau_logdvol - s.logvol))
        s.logvol -= cur_dlv
        out[e] = s.logvol
        cur[dead_id[e]] = repl[e]
        if s.plateau_mode:
            s.plateau_counter -= 1
            if s.plateau_counter == 0:
                s.plateau_mode = False
    if not s.plateau_mode:
        rel = np.log(1. - (np.arange(N) + 1.) / (N + 1.))
    else:
        rel = np.log1p(-((1 + np.arange(s.plateau_counter)) * np.exp(s.plateau_logdvol - s.logvol)))
        nrest = N - s.plateau_counter
        rel = np.concatenate([rel, rel[-1] + np.log1p(-(1 + np.arange(nrest)) / (nrest + 1))])
    return np.concatenate([out, rel + s.logvol])


# ---------------------------------------------------------------------------
# streams of the resident loop's bootstrap replicas (dynesty_amd/csrc/boot.hip)
# ---------------------------------------------------------------------------
_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_M128 = (1 << 128) - 1


def boot_generator(ent, b):
    """The NumPy Generator replica `b` of a rebuild resamples with: PCG64 seeded by
    pcg_setseq_128_srandom_r(initstate = (ent[0], ent[1] + b), initseq = (ent[2], ent[3] + 2 b)), the four
    64-bit words `ent` being what the run drew for this rebuild (high word first)."""
    import numpy as np
    e = [int(x) for x in ent]
    m64 = (1 << 64) - 1
    initstate = (e[0] << 64) | ((e[1] + b) & m64)
    initseq = (e[2] << 64) | ((e[3] + 2 * b) & m64)
    inc = ((initseq << 1) | 1) & _M128
    state = inc  # 0 * MULT + inc
    state = (state + initstate) & _M128
    state = (state * _PCG_MULT + inc) & _M128
    bg = np.random.PCG64()
    bg.state = {'bit_generator': 'PCG64', 'state': {'state': state, 'inc': inc}, 'has_uint32': 0, 'uinteger': 0}
    return np.random.Generator(bg)


def boot_expand(points, ent, bootstrap, multi):
    """max over the replicas of _ellipsoid_bootstrap_expand (ref: bounding.py:381-400 / 688-703, 1619-1648)
    with the replica streams of boot_generator."""
    from . import bounding_ref as B
    return max(B.bootstrap_expand(multi, points, boot_generator(ent, b)) for b in range(bootstrap))
