"""Batched execution of dynesty's proposal samplers on the device.

Each ``run_*`` function takes the list of ``SamplerArgument`` tuples that
``InternalSampler.prepare_sampler`` built for one queue fill
(reference sampler.py:676-717, internal_samplers.py:111-159) and returns the
list of ``SamplerReturn``-shaped results, one device launch for the whole
queue.  All walkers of a fill share ``loglstar`` and ``scale``
(sampler.py:709-716).

Random streams: every argument carries ``rseed`` -- the sampler's own
``numpy.random.Generator`` when ``queue_size == 1`` or a ``SeedSequence`` child
otherwise (sampler.py:696-699).  The device continues exactly those PCG64
streams: Generators are read and written back, SeedSequence children are hashed
on the device.
"""
import warnings
from collections import namedtuple

import numpy as np

from . import _lib
from .backend import get_backend

# same field names as internal_samplers.py:28-31
SamplerReturn = namedtuple('SamplerReturn', [
    'u', 'v', 'logl', 'ncalls', 'evaluation_history', 'tuning_info',
    'proposal_stats'
])


# same field names as utils.py:23-24
SamplerHistoryItem = namedtuple('SamplerHistoryItem', ['u', 'v', 'logl'])


class NoDeviceProblem(NotImplementedError):
    pass


def _wants_history(arg):
    """utils.LogLikelihood(save_evaluation_history=True): the sampler must return every evaluated point."""
    return bool(getattr(arg.loglikelihood, 'save_evaluation_history', False))


def _problem_of(arg):
    prob = arg.kwargs.get('problem')
    if prob is None:
        raise NoDeviceProblem(
            "this sampler was built without problem=<dynesty_amd.problems."
            "Problem>: the device proposal kernels evaluate the prior transform "
            "and the log-likelihood in-kernel and need their device twin. "
            "(A lock-step path for arbitrary Python callbacks is not built.)")
    return prob


class _Streams:
    """Per-walker PCG64 states for a list of rseeds + write-back."""

    def __init__(self, seeds):
        self.seeds = seeds if isinstance(seeds, list) else list(seeds)
        k = len(self.seeds)
        be = get_backend()
        first = self.seeds[0]
        self.generators = None
        if isinstance(first, np.random.Generator):
            self.generators = self.seeds
            self.states = np.array([
                _lib.pcg_state_words(g.bit_generator) for g in self.seeds
            ], dtype=np.uint64)
            return
        # SeedSequence children of one parent, consecutive spawn keys: hash on
        # the device (utils.py:1002-1009 spawns exactly this)
        ent = first.entropy
        key0 = first.spawn_key
        consecutive = len(key0) == 1 and first.pool_size == 4
        if consecutive:
            base = key0[0]
            # spawn() hands every child the parent's entropy object itself: identity is the cheap test,
            # equality the fallback (a fill is 512+ seeds; np.array_equal per seed was 4 ms of a fill)
            for i, sq in enumerate(self.seeds):
                if sq.spawn_key != (base + i,) or (
                        sq.entropy is not ent and not np.array_equal(
                            np.atleast_1d(sq.entropy), np.atleast_1d(ent))):
                    consecutive = False
                    break
        if consecutive:
            self.states = be.seed_children(np.atleast_1d(ent), key0[0], k)
        else:
            self.states = np.array([
                _lib.pcg_state_words(np.random.PCG64(sq)) for sq in self.seeds
            ], dtype=np.uint64)

    def write_back(self, rng_out):
        if self.generators is not None:
            for g, st in zip(self.generators, rng_out):
                _lib.set_pcg_state_words(g.bit_generator, st)


def _frames(args):
    """Unique proposal frames + index per walker.  Frames are deduplicated by object
    (HipMultiEllipsoid.get_random_axes hands out one cached view per ellipsoid), then by
    the memory they view (the reference's MultiEllipsoid returns a fresh ndarray view of
    `ells[i].axes` per call: same data pointer, different `id`), then by value, so a fill
    uploads one frame per ellipsoid, not one per walker."""
    uniq, by_id, by_mem, by_val = [], {}, {}, {}
    idx = np.empty(len(args), dtype=np.int32)
    for i, a in enumerate(args):
        ax = a.axes
        j = by_id.get(id(ax))
        if j is None:
            if isinstance(ax, np.ndarray):
                key = (ax.__array_interface__['data'][0], ax.shape, ax.strides, ax.dtype.str)
            else:
                key = ('id', id(ax))
            j = by_mem.get(key)
            if j is None:
                arr = np.ascontiguousarray(ax, dtype=np.float64)
                vkey = (arr.shape, arr.tobytes())
                j = by_val.get(vkey)
                if j is None:
                    j = len(uniq)
                    by_val[vkey] = j
                    uniq.append(arr)
                by_mem[key] = j
            by_id[id(ax)] = j
        idx[i] = j
    return np.stack(uniq), (None if len(uniq) == 1 else idx)


_NO_HISTORY = ()  # SamplerReturn.evaluation_history of the fused samplers: read-only, shared (sampler.py:748 only extends FROM it)


def _start_points(args):
    """(k, ndim) float64 of the arguments' start points in one copy."""
    return np.array([a.u for a in args], dtype=np.float64)


def _returns(u, v, logl, ncalls, tuning, stats):
    """One SamplerReturn per walker from whole-fill arrays / lists without a Python-level loop body: rows of u and
    v are views of the output arrays, scalars are converted by tolist() in one pass."""
    k = len(u)
    if not isinstance(ncalls, list):
        ncalls = [ncalls] * k
    return list(map(SamplerReturn._make,
                    zip(list(u), list(v), logl.tolist(), ncalls, [_NO_HISTORY] * k, tuning, stats)))


def _bc_flags(kwargs, ndim):
    periodic, reflective = kwargs.get('periodic'), kwargs.get('reflective')
    if periodic is None and reflective is None:
        return None
    bc = np.zeros(ndim, dtype=np.int8)
    if periodic is not None:
        bc[np.asarray(periodic)] = _lib.BC_PERIODIC
    if reflective is not None:
        bc[np.asarray(reflective)] = _lib.BC_REFLECT
    return bc


def _run_rwalk_lockstep(args, history=False):
    """rwalk with an arbitrary Python likelihood: the device proposes one
    propose_ball_point per walker per step (dh_rwalk_propose), the host
    evaluates the user's callbacks and applies generic_random_walk's accept
    rule (internal_samplers.py:925-975).  Same streams, same counters.
    history=True: every evaluated proposal is kept as a SamplerHistoryItem
    (internal_samplers.py:960-961), in the walker's order."""
    a0 = args[0]
    kw = a0.kwargs
    k = len(args)
    u = np.array([a.u for a in args], dtype=np.float64)
    ndim = u.shape[1]
    axes, idx = _frames(args)
    streams = _Streams([a.rseed for a in args])
    states = streams.states
    be = get_backend()
    bc = _bc_flags(kw, ndim)
    walks = int(kw['walks'])
    nacc = np.zeros(k, dtype=np.int64)
    nrej = np.zeros(k, dtype=np.int64)
    v = [None] * k
    logl = [None] * k
    hist = [[] for _ in range(k)] if history else None
    for _ in range(walks):
        up, inside, states = be.rwalk_propose(u, axes, a0.scale, states,
                                              axes_idx=idx,
                                              ncdim=axes.shape[1], bc=bc)
        for i in range(k):
            if not inside[i]:
                nrej[i] += 1
                continue
            vi = a0.prior_transform(np.array(up[i]))
            li = a0.loglikelihood(np.asarray(vi))
            if history:
                hist[i].append(SamplerHistoryItem(u=np.array(up[i]), v=vi, logl=li))
            if li > a0.loglstar:
                u[i], v[i], logl[i] = up[i], vi, li
                nacc[i] += 1
            else:
                nrej[i] += 1
    streams.write_back(states)
    res = []
    for i in range(k):
        if nacc[i] == 0:
            v[i] = a0.prior_transform(np.array(u[i]))
            logl[i] = a0.loglikelihood(np.asarray(v[i]))
        res.append(SamplerReturn(
            u=u[i].copy(), v=v[i], logl=logl[i], ncalls=walks,
            evaluation_history=hist[i] if history else [],
            tuning_info={'accept': int(nacc[i]), 'reject': int(nrej[i]),
                         'scale': a0.scale},
            proposal_stats=dict(n_accept=int(nacc[i]), n_reject=int(nrej[i]))))
    return res


def stored_logl_of(points, nested_sampler):
    """ln L that the run holds for each start point of a fill, or None where it cannot be told.  A start point is a
    copy of a live point (sampler.py:676-700, propose_live), so its row is found among `nested_sampler.live_u` bit for
    bit and `live_logl` has its value.  (Duplicates of a row -- an unmoved walker's earlier return -- carry the same
    stored value, so any match serves.)"""
    live_u = getattr(nested_sampler, 'live_u', None)
    live_logl = getattr(nested_sampler, 'live_logl', None)
    if live_u is None or live_logl is None or len(points) == 0:
        return None
    live_u = np.asarray(live_u)
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim != 2 or live_u.ndim != 2 or pts.shape[1] != live_u.shape[1]:
        return None
    live_logl = np.asarray(live_logl, dtype=np.float64)
    order = np.argsort(live_u[:, 0], kind='stable')
    col = live_u[order, 0]
    out = np.full(len(pts), np.nan)
    lo = np.searchsorted(col, pts[:, 0], side='left')
    hi = np.searchsorted(col, pts[:, 0], side='right')
    has = hi > lo
    cand = order[np.minimum(lo, len(order) - 1)]
    hit = has & np.all(live_u[cand] == pts, axis=1)  # the common case: one row with that first coordinate
    out[hit] = live_logl[cand[hit]]
    for i in np.nonzero(has & ~hit)[0]:  # several rows share the first coordinate: look at each
        for j in order[lo[i]:hi[i]]:
            if np.array_equal(live_u[j], pts[i]):
                out[i] = live_logl[j]
                break
    return out


def run_rwalk(args):
    """RWalkSampler.sample over a queue (internal_samplers.py:504-561).

    A walker that accepted no step returns its start point (internal_samplers.py:546-553).  The reference then
    evaluates ln L of that point again -- the same function of the same bits as when the point entered the live set,
    hence the stored value exactly, an exact tie that `np.argmin` breaks by slot.  A device kernel's evaluation of the
    point can differ from the stored value in the last bit (the point may have been born in another kernel, or on the
    host), which would order the two copies by rounding instead: when the fill's arguments carry the stored values
    (`kwargs['logl0']`, put there by HipRWalkSampler.prepare_sampler) an unmoved walker hands back THAT value, as the
    resident loop does."""
    args = list(args)
    if not args:
        return []
    a0 = args[0]
    if a0.kwargs.get('problem') is None or _wants_history(a0):
        return _run_rwalk_lockstep(args, history=_wants_history(a0))
    prob = _problem_of(a0)
    kw = a0.kwargs
    u0 = _start_points(args)
    axes, idx = _frames(args)
    streams = _Streams([a.rseed for a in args])
    out = get_backend().rwalk_batch(
        prob, u0, axes, a0.scale, a0.loglstar, kw['walks'], streams.states,
        axes_idx=idx, ncdim=axes.shape[1], bc=_bc_flags(kw, prob.ndim))
    streams.write_back(out["rng_out"])
    walks = int(kw['walks'])
    scale = a0.scale
    acc, rej = out["accept"].tolist(), out["reject"].tolist()
    if 'logl0' in kw:
        logl0 = np.array([a.kwargs.get('logl0', np.nan) for a in args], dtype=np.float64)
        unmoved = (np.asarray(out["accept"]) == 0) & np.isfinite(logl0) & np.all(np.asarray(out["u"]) == u0, axis=1)
        if unmoved.any():
            out["logl"] = np.where(unmoved, logl0, out["logl"])
    return _returns(out["u"], out["v"], out["logl"], walks,
                    [{'accept': na, 'reject': nr, 'scale': scale} for na, nr in zip(acc, rej)],
                    [{'n_accept': na, 'n_reject': nr} for na, nr in zip(acc, rej)])


class _UniformFeed:
    """Generator.random() of k walker streams, served from the device's
    uncommitted lookahead (dh_slice_feed); `begin` commits what was used."""

    def __init__(self, be, ndim, states6, nlook):
        self.be, self.ndim, self.nlook = be, ndim, int(nlook)
        self.states = np.array(states6, dtype=np.uint64).reshape(-1, 6)
        k = self.states.shape[0]
        self.pos = np.zeros(k, dtype=np.int32)
        self.look = np.empty((k, 0))

    def begin(self, kind, **kw):
        self.states, out, self.look = self.be.slice_feed(
            kind, self.ndim, self.states, consumed=self.pos, nlook=self.nlook,
            **kw)
        self.pos[:] = 0
        return out

    def random(self, i):
        if self.pos[i] == self.look.shape[1]:  # lookahead used up: refill walker i
            st, _, lk = self.be.slice_feed(
                'advance', self.ndim, self.states[i:i + 1],
                consumed=self.pos[i:i + 1], nlook=self.nlook)
            self.states[i] = st[0]
            self.look[i] = lk[0]
            self.pos[i] = 0
        x = self.look[i, self.pos[i]]
        self.pos[i] += 1
        return float(x)

    def commit(self):
        self.states, _, _ = self.be.slice_feed('advance', self.ndim, self.states,
                                               consumed=self.pos, nlook=0)
        self.pos[:] = 0
        return self.states


def _unitcheck(u, nonperiodic):
    # utils.unitcheck: periodic coordinates may leave the cube by half a period
    if nonperiodic is None:
        return bool(np.all(u > 0.) and np.all(u < 1.))
    nonp = np.asarray(nonperiodic, dtype=bool)
    return bool(np.all(u[nonp] > 0.) and np.all(u[nonp] < 1.)
                and np.all(u[~nonp] > -0.5) and np.all(u[~nonp] < 1.5))


def _slice_step_host(u, direction, nonperiodic, loglstar, ptform, loglike,
                     doubling, rand, hist=None):
    """generic_slice_step (internal_samplers.py:1076-1206) with the uniforms
    of the walker's stream handed in by `rand()`.  Returns (u, v, logl, nc,
    n_expand, n_contract, expansion_warning)."""
    n = len(u)
    nc = n_expand = n_contract = 0
    rand0 = rand()
    dirlen = np.linalg.norm(direction)
    maxlen = np.sqrt(n) / 2.
    direction = direction / (dirlen / maxlen if dirlen > maxlen else 1)

    def F(x):
        nonlocal nc
        u_new = u + x * direction
        nc += 1
        if _unitcheck(u_new, nonperiodic):
            v_new = ptform(u_new)
            l_new = loglike(v_new)
            if hist is not None:  # internal_samplers.py:1118-1119
                hist.append(SamplerHistoryItem(u=u_new, v=v_new, logl=l_new))
            return u_new, v_new, l_new
        return u_new, None, -np.inf

    left, right = -rand0, 1 - rand0
    f_l, f_r = F(left)[2], F(right)[2]
    warn = False
    if not doubling:
        while f_l > loglstar:
            left -= 1
            f_l = F(left)[2]
            n_expand += 1
        while f_r > loglstar:
            right += 1
            f_r = F(right)[2]
            n_expand += 1
        if n_expand > 1000:
            warn = True
            warnings.warn('The slice sample interval was expanded more '
                          'than 1000 times')
    else:
        K = 1
        while f_l > loglstar or f_r > loglstar:
            if rand() < 0.5:
                left -= (right - left)
                f_l = F(left)[2]
            else:
                right += (right - left)
                f_r = F(right)[2]
            n_expand += K
            K *= 2
        L, R, fL, fR = left, right, f_l, f_r
    while True:
        x = left + rand() * (right - left)
        u_prop, v_prop, f = F(x)
        n_contract += 1
        ok = f > loglstar
        if ok and doubling:  # Neal (2003) algorithm 6, w = 1, x0 = 0
            lhat, rhat, f_lhat, f_rhat, far = L, R, fL, fR, False
            while rhat - lhat > 1.1:
                M = (lhat + rhat) / 2.
                if (0 < M <= x) or (x < M <= 0):
                    far = True
                if x < M:
                    rhat = M
                    f_rhat = F(rhat)[2]
                else:
                    lhat = M
                    f_lhat = F(lhat)[2]
                if far and loglstar >= f_lhat and loglstar >= f_rhat:
                    ok = False
                    break
        if ok:
            return u_prop, v_prop, f, nc, n_expand, n_contract, warn
        if x < 0:
            left = x
        elif x > 0:
            right = x
        else:
            raise RuntimeError(
                "Slice sampler has failed to find a valid point. Some useful "
                f"output quantities:\nu: {u}\nnstep_left: {left}\n"
                f"nstep_right: {right}\nu_prop: {u_prop}\nloglstar: {loglstar}\n"
                f"logl_prop: {f}\ndirection: {direction}\n")


def _run_slice_lockstep(args, principal, history=False):
    """RSliceSampler / SliceSampler with an arbitrary Python likelihood: per
    slice ONE device call hands every walker its direction (or shuffled axis
    order) and the uniforms that follow in its stream; the host runs
    generic_slice_step's state machine around the user's callbacks
    (internal_samplers.py:593-855).  Same streams, same counters."""
    a0 = args[0]
    kw = a0.kwargs
    k = len(args)
    u = np.array([a.u for a in args], dtype=np.float64)
    ndim = u.shape[1]
    nonperiodic = kw.get('nonperiodic', None)
    doubling = [bool(kw.get('slice_doubling', False))] * k
    slices = int(kw['slices'])
    axes, idx = _frames(args)
    streams = _Streams([a.rseed for a in args])
    st6 = np.concatenate([np.array(streams.states, dtype=np.uint64).reshape(k, 4),
                          np.zeros((k, 2), dtype=np.uint64)], axis=1)
    if streams.generators is not None:
        st6 = np.array([_lib.pcg_state6(g.bit_generator)
                        for g in streams.generators], dtype=np.uint64)
    feed = _UniformFeed(get_backend(), ndim, st6,
                        nlook=(max(64, 8 * ndim) if principal else 64))
    nc = np.zeros(k, dtype=np.int64)
    ne = np.zeros(k, dtype=np.int64)
    nt = np.zeros(k, dtype=np.int64)
    warn_set = [False] * k
    v = [None] * k
    logl = [None] * k
    hist = [[] for _ in range(k)] if history else None
    # axes[:, i] is the i-th principal axis (internal_samplers.py:660-663)
    scaled = [a0.scale * np.asarray(fr).T for fr in axes] if principal else None
    for _ in range(slices):
        if principal:
            lead = feed.begin('shuffle')
        else:
            lead = feed.begin('direction', axes=axes, axes_idx=idx,
                              scale=a0.scale)
        for i in range(k):
            rand = (lambda i=i: feed.random(i))
            steps = ([scaled[0 if idx is None else idx[i]][j] for j in lead[i]]
                     if principal else [lead[i]])
            for direction in steps:
                (u[i], v[i], logl[i], c1, e1, t1, w1) = _slice_step_host(
                    u[i], direction, nonperiodic, a0.loglstar,
                    a0.prior_transform, a0.loglikelihood, doubling[i], rand,
                    hist[i] if history else None)
                nc[i] += c1
                ne[i] += e1
                nt[i] += t1
                if w1 and not doubling[i]:
                    doubling[i] = True
                    warn_set[i] = True
                    warnings.warn('Enabling doubling strategy of slice '
                                  'sampling from Neal(2003)')
    st6 = feed.commit()
    if streams.generators is not None:
        for g, st in zip(streams.generators, st6):
            _lib.set_pcg_state6(g.bit_generator, st)
    res = []
    for i in range(k):
        res.append(SamplerReturn(
            u=u[i].copy(), v=v[i], logl=logl[i], ncalls=int(nc[i]),
            evaluation_history=hist[i] if history else [],
            tuning_info={'n_expand': int(ne[i]), 'n_contract': int(nt[i]),
                         'expansion_warning_set': warn_set[i]},
            proposal_stats=dict(n_expand=int(ne[i]), n_contract=int(nt[i]))))
    return res


def _run_slice(args, principal):
    args = list(args)
    if not args:
        return []
    a0 = args[0]
    nonp = a0.kwargs.get('nonperiodic', None)
    if a0.kwargs.get('problem') is None or _wants_history(a0) or (
            nonp is not None and not np.all(nonp)):
        # arbitrary Python likelihood, an evaluation history wanted, or periodic
        # coordinates (the fused slice kernels implement the plain unit-cube check only)
        return _run_slice_lockstep(args, principal, history=_wants_history(a0))
    prob = _problem_of(a0)
    kw = a0.kwargs
    u0 = _start_points(args)
    axes, idx = _frames(args)
    streams = _Streams([a.rseed for a in args])
    out = get_backend().slice_batch(
        prob, u0, axes, a0.scale, a0.loglstar, kw['slices'], streams.states,
        principal=principal, doubling=bool(kw.get('slice_doubling', False)),
        axes_idx=idx)
    streams.write_back(out["rng_out"])
    ne, nt = out["n_expand"].tolist(), out["n_contract"].tolist()
    ws = np.asarray(out["expansion_warning_set"]).astype(bool).tolist()
    return _returns(out["u"], out["v"], out["logl"], np.asarray(out["ncalls"]).tolist(),
                    [{'n_expand': e, 'n_contract': t, 'expansion_warning_set': w} for e, t, w in zip(ne, nt, ws)],
                    [{'n_expand': e, 'n_contract': t} for e, t in zip(ne, nt)])


def run_rslice(args):
    """RSliceSampler.sample over a queue (internal_samplers.py:745-855)."""
    return _run_slice(args, principal=False)


def run_slice(args):
    """SliceSampler.sample over a queue (internal_samplers.py:593-709)."""
    return _run_slice(args, principal=True)


def bound_arrays(bound):
    """(ctrs, axes, ams, logvol_ells) of an ellipsoidal bound -- ours or the
    reference's (duck-typed: bounding.py:201-240, 440-476)."""
    if hasattr(bound, 'ctrs') and hasattr(bound, 'logvol_ells'):
        axes = getattr(bound, 'axes_ells', None)
        if axes is None:
            axes = np.array([e.axes for e in bound.ells])
        return (np.asarray(bound.ctrs), np.asarray(axes), np.asarray(bound.ams),
                np.asarray(bound.logvol_ells))
    if hasattr(bound, 'ctr') and hasattr(bound, 'axes'):
        return (np.asarray(bound.ctr)[None], np.asarray(bound.axes)[None],
                np.asarray(bound.am)[None], np.array([bound.logvol]))
    raise TypeError(
        f"{type(bound).__name__} is not an ellipsoidal bound: the device "
        "uniform sampler supports Ellipsoid / MultiEllipsoid bounds")


def friends_kind(bound):
    """'balls' / 'cubes' for a RadFriends / SupFriends bound -- ours or the
    reference's (duck-typed on axes_inv + need_centers; the reference's two
    classes differ in their `within` norm, told apart by name)."""
    if not (hasattr(bound, 'axes_inv') and getattr(bound, 'need_centers', False)):
        return None
    kind = getattr(bound, 'kind', None)
    if kind is None:
        kind = 'cubes' if 'Sup' in type(bound).__name__ else 'balls'
    return kind


def _run_unif_lockstep(args, history=False):
    """UniformBoundSampler / UnitCubeSampler semantics with an arbitrary Python
    likelihood: per round the device hands every unfinished walker the next
    candidate of its stream that lies in the bound and passes unitcheck
    (dh_unif_batch with problem = -1); the host evaluates the user's callbacks
    and keeps the first candidate with logl > loglstar
    (internal_samplers.py:304-333).  Same streams and call counts as the fused
    kernel."""
    a0 = args[0]
    kw = a0.kwargs
    k = len(args)
    ndim = int(kw['ndim'])
    bound = kw['bound']
    streams = _Streams([a.rseed for a in args])
    states = np.array(streams.states, dtype=np.uint64).reshape(k, 4)
    bc = None
    nonb = kw.get('nonbounded')
    if nonb is not None:
        bc = np.where(np.asarray(nonb), _lib.BC_HARD,
                      _lib.BC_PERIODIC).astype(np.int8)
    fkind = friends_kind(bound)
    pk = {}
    if fkind is not None:
        if kw['n_cluster'] != ndim:
            raise ValueError("balls / cubes bounds need ncdim == ndim")
        pk = dict(friends=(fkind, np.asarray(bound.ctrs), np.real(bound.axes),
                           np.real(bound.axes_inv)))
        # + {has_uint32, uinteger}: integers(n) draws from the buffered 32-bit half
        states = np.concatenate([states, np.zeros((k, 2), dtype=np.uint64)], axis=1)
    else:
        ctrs, axes, ams, lvs = bound_arrays(bound)
        pk = dict(ctrs=ctrs, axes=axes, ams=ams, logvol_ells=lvs,
                  ncdim=kw['n_cluster'])
    be = get_backend()
    todo = np.arange(k)
    res = [None] * k
    nc = np.zeros(k, dtype=np.int64)
    hist = [[] for _ in range(k)] if history else None
    while len(todo):
        up, out = be.unif_propose(ndim, states[todo], bc=bc, **pk)
        states[todo] = out
        keep = []
        for j, i in enumerate(todo):
            vi = a0.prior_transform(np.array(up[j]))
            li = a0.loglikelihood(np.asarray(vi))
            nc[i] += 1
            if history:  # internal_samplers.py:330
                hist[i].append(SamplerHistoryItem(u=up[j].copy(), v=vi, logl=li))
            if li > a0.loglstar:
                res[i] = SamplerReturn(u=up[j].copy(), v=vi, logl=li,
                                       ncalls=int(nc[i]),
                                       evaluation_history=hist[i] if history else [],
                                       tuning_info=None,
                                       proposal_stats={'n_proposals': 0})
            else:
                keep.append(i)
        todo = np.array(keep, dtype=np.int64)
    streams.write_back(states[:, :4])
    return res


def run_unif(args):
    """UniformBoundSampler.sample over a queue (internal_samplers.py:243-340)."""
    args = list(args)
    if not args:
        return []
    a0 = args[0]
    if a0.kwargs.get('problem') is None or _wants_history(a0):
        return _run_unif_lockstep(args, history=_wants_history(a0))
    prob = _problem_of(a0)
    kw = a0.kwargs
    bound = kw['bound']
    streams = _Streams([a.rseed for a in args])
    bc = None
    nonb = kw.get('nonbounded')
    if nonb is not None:
        bc = np.where(np.asarray(nonb), _lib.BC_HARD,
                      _lib.BC_PERIODIC).astype(np.int8)
    fkind = friends_kind(bound)
    if fkind is not None:
        if kw['n_cluster'] != kw['ndim']:
            raise ValueError("balls / cubes bounds need ncdim == ndim (their "
                             "centres are the full live points)")
        out = get_backend().unif_friends_batch(
            prob, a0.loglstar, streams.states, np.asarray(bound.ctrs), fkind,
            np.real(bound.axes), np.real(bound.axes_inv), bc=bc)
    else:
        ctrs, axes, ams, lvs = bound_arrays(bound)
        out = get_backend().unif_batch(prob, a0.loglstar, streams.states,
                                       ctrs=ctrs, axes=axes, ams=ams,
                                       logvol_ells=lvs, ncdim=kw['n_cluster'],
                                       bc=bc)
    streams.write_back(out["rng_out"])
    k = len(args)
    return _returns(out["u"], out["v"], out["logl"], np.asarray(out["ncalls"]).tolist(), [None] * k,
                    [{'n_proposals': 0} for _ in range(k)])


def _device_problem_of(arg):
    """The dynesty_amd.problems.Problem behind an argument's callbacks, or None: dynesty wraps the user's functions
    (utils.LogLikelihood(.loglikelihood), _function_wrapper(.func)); underneath must sit the bound methods
    `prob.loglikelihood` and `prob.prior_transform` of ONE Problem (the object that has a device twin)."""
    def owner(fn):
        """(Problem, method name) at the bottom of a wrapper chain, or (None, '').  Every level on the way must be a
        plain pass-through: a wrapper that holds extra positional / keyword arguments (utils._function_wrapper's
        args / kwargs = the sampler's logl_args / ptform_args) or returns blobs (utils.LogLikelihood.blob) computes
        something the device twin does not."""
        for _ in range(6):
            if getattr(fn, 'args', None) or getattr(fn, 'kwargs', None) or getattr(fn, 'blob', False):
                return None, ''
            own = getattr(fn, '__self__', None)
            if own is not None and hasattr(own, 'device_spec'):
                return own, getattr(fn, '__name__', '')
            nxt = getattr(fn, 'loglikelihood', None) or getattr(fn, 'func', None)
            if nxt is None or nxt is fn:
                return None, ''
            fn = nxt
        return None, ''
    pl, nl = owner(arg.loglikelihood)
    pp, npt = owner(arg.prior_transform)
    if pl is None or pl is not pp or nl != 'loglikelihood' or npt != 'prior_transform':
        return None
    return pl


def run_unitcube(args):
    """UnitCubeSampler.sample over a queue (internal_samplers.py:364-441) -- dynesty's own sampler of the phase before
    the first bound, which it builds itself (sampler.py: the internal sampler until update_bound_if_needed switches).
    HipBatchPool.map recognises it; when the run's callbacks are a device Problem's the whole queue is one launch
    (dh_unif_batch without a bound: same streams, same call counts), else None (the caller maps it serially).  At C2
    the serial form was 0.69 s of a 6 s tap-B run: 16 fills x 512 walkers of the reference's Python."""
    args = args if isinstance(args, list) else list(args)
    if not args:
        return []
    a0 = args[0]
    if _wants_history(a0):
        return None
    prob = _device_problem_of(a0)
    if prob is None or a0.kwargs.get('ndim') != prob.ndim:
        return None
    streams = _Streams([a.rseed for a in args])
    out = get_backend().unif_batch(prob, a0.loglstar, streams.states)
    streams.write_back(out["rng_out"])
    nc = np.asarray(out["ncalls"]).tolist()
    return _returns(out["u"], out["v"], out["logl"], nc, [None] * len(args), [{'n_proposals': c} for c in nc])


def batched(runner):
    """Wrap a queue runner as the static per-argument ``sample`` dynesty
    expects; ``HipBatchPool.map`` finds the runner on ``_dynhip_batch``."""

    def checked(args):
        # save_evaluation_history=True (utils.LogLikelihood, utils.py:120-262; internal_samplers.py:311, 426, 663)
        # wants every point a sampler evaluated.  The fused kernels keep those in registers, so such a run takes the
        # lock-step runners instead -- the device proposes (same streams, same counters), the host evaluates the
        # caller's prior transform and likelihood and keeps the history: the opt-in slow path (see _wants_history)
        return runner(args)

    def sample(arg):
        return checked([arg])[0]

    sample._dynhip_batch = checked
    sample.__doc__ = runner.__doc__
    return sample
