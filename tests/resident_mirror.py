"""Test infrastructure: one run of the device-resident loop (csrc/ns.hip, dh_ns_ensemble) restated on the host, event
for event.

What is restated here is the loop's CONTROL -- what Sampler.sample / _new_point / _fill_queue / propose_live /
update_bound_if_needed do between the numerical steps (sampler.py:469-489, 625-778, 1070-1195), in the resident loop's
own protocol for the random choices (ns_init / ns_select: SeedSequence children per run and per initial point, one PCG64
per queue entry seeded from four words of the run's generator).  The numerical steps themselves are the library's own
entry points, called one at a time through the host API (bound.update -> dh_rebuild, scale_to_logvol, contains_many,
dh_unif_batch for the unit-cube phase, dh_rwalk_batch, dh_ns_consume), so a mirrored run and the same run inside
dh_ns_ensemble take the same proposals and must agree death for death: slots, replacement sources, iteration and
call counts, the number of bound updates, ln Z.  (Scalars that pass through libm on one side and ocml on the other --
the tuned scale, the enlargement target -- can differ in the last bit: coordinates are compared to 1e-12, not bit for
bit.)  forced = 'late' | 'exact' selects the loop's two forms of the forced bound update (DH_NS_OPT_FORCED_EXACT).
"""
import math

import numpy as np

M64 = (1 << 64) - 1
M128 = (1 << 128) - 1
PCG_MULT = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645


class Pcg:
    """csrc/rng_pcg64.h: Pcg64 (numpy's PCG64: 128-bit LCG, XSL-RR output, buffered 32-bit halves)."""

    def __init__(self, words=None):
        self.state = self.inc = 0
        self.has32, self.buf32 = 0, 0
        if words is not None:
            self.load(words)

    def load(self, w):
        self.state = (int(w[0]) << 64) | int(w[1])
        self.inc = (int(w[2]) << 64) | int(w[3])
        self.has32 = 0

    def words(self):
        return np.array([self.state >> 64, self.state & M64, self.inc >> 64, self.inc & M64], dtype=np.uint64)

    def step(self):
        self.state = (self.state * PCG_MULT + self.inc) & M128

    def next64(self):
        self.step()
        hi, lo = self.state >> 64, self.state & M64
        x, r = hi ^ lo, hi >> 58
        return ((x >> r) | (x << ((64 - r) & 63))) & M64

    def next32(self):
        if self.has32:
            self.has32 = 0
            return self.buf32
        v = self.next64()
        self.has32, self.buf32 = 1, v >> 32
        return v & 0xFFFFFFFF

    def next_double(self):
        return (self.next64() >> 11) * (1.0 / 9007199254740992.0)

    def interval(self, mx):  # random_interval(max): masked rejection on the buffered 32-bit stream
        if mx == 0:
            return 0
        mask = (1 << mx.bit_length()) - 1
        while True:
            v = (self.next32() if mx <= 0xFFFFFFFF else self.next64()) & mask
            if v <= mx:
                return v

    def seed(self, initstate, initseq):  # pcg_setseq_128_srandom_r
        self.state = 0
        self.inc = ((initseq << 1) | 1) & M128
        self.step()
        self.state = (self.state + initstate) & M128
        self.step()
        self.has32 = 0


def child_words(entropy, child):
    """seed_from_child: PCG64(SeedSequence(entropy, spawn_key=(child,))) as four state words."""
    bg = np.random.PCG64(np.random.SeedSequence(list(entropy), spawn_key=(int(child),)))
    st = bg.state["state"]
    s, i = int(st["state"]), int(st["inc"])
    return np.array([s >> 64, s & M64, i >> 64, i & M64], dtype=np.uint64)


def mirror_run(ctx, prob, nlive, K, walks, bound, entropy, run, dlogz, enlarge=1.25, forced="exact", first_run=0,
               max_fills=100000, sample="rwalk", bc=None, bootstrap=0, update_interval=None, first_update=None):
    """The run with global index first_run + run of ns_ensemble(prob, ..., rebuild_every=1); sample = 'rwalk' |
    'rslice' | 'slice' (`walks` is then the number of slices), bc = DH_BC_* flags per dimension or None."""
    from dynesty_amd import backend, bounding
    backend.set_backend(ctx)
    try:
        return _mirror(ctx, prob, nlive, K, walks, bound, entropy, first_run + run, dlogz, enlarge, forced, max_fills,
                       sample, bc, bootstrap, update_interval, first_update)
    finally:
        backend.set_backend(None)


def _mirror(ctx, prob, N, K, walks, bound, entropy, grun, dlogz, enlarge, forced, max_fills, sample, bc, bootstrap,
            upd, first):
    from dynesty_amd import bounding
    D = prob.ndim
    # ---- ns_init: every initial point its own child stream, the run's generator child 0x80000000 + run ----
    live_u = np.empty((N, D))
    for i in range(N):
        g = Pcg(child_words(entropy, grun * N + i))
        live_u[i] = [g.next_double() for _ in range(D)]
    rg = Pcg(child_words(entropy, 0x80000000 + grun))
    live_v, live_logl = ctx.problem_eval(prob, live_u)
    live_l2 = np.ascontiguousarray(live_logl, dtype=np.float64)[None, :]
    live_logl = live_l2[0]
    state = np.array([[0., -1.e300, 0., 0., -1.e300, 0., float(N), 0.]])
    plateau = np.zeros((1, 2))
    live_it2 = np.zeros((1, N), dtype=np.int32)
    loglstar = float(live_logl.min())
    facc = min(1.0, max(1.0 / max(walks, 2), 0.5))
    # internal_samplers.py:88-94 (unif: 1), 495-502 (rwalk: walks), 582 (slice: slices x ndim), 737 (rslice: slices)
    update_interval = N if sample == "unif" else walks * N * (D if sample == "slice" else 1)
    doubling = False
    first_ncall, first_eff = 2 * N, 10.0
    if upd is not None:  # dynesty.py:213-234: a float is a multiple of nlive, an int a number of calls
        update_interval = max(1, round(upd * N)) if isinstance(upd, float) else int(upd)
    if first:
        first_ncall, first_eff = first.get("min_ncall", first_ncall), first.get("min_eff", first_eff)
    cube, scale, nbound, ncall_last, force = True, 1.0, 0, 0, False
    carry = 0  # calls of the entries popped since the last death (sampler.py:739-747: _new_point's ncall_accum)
    undo = None  # (slot, its content before the replacement) when the previous fill's last entry made the last death
    bnd = None
    ev = dict(dead_logl=[], dead_slot=[], dead_src=[], fill_of_death=[], forced_fills=[], rebuild_fills=[])

    def rebuild():
        nonlocal bnd, nbound
        if bnd is None:
            bnd = (bounding.HipMultiEllipsoid if bound == "multi" else bounding.HipEllipsoid)(D)
        bnd.update(live_u)
        if bootstrap > 0:
            # bound.update(points, bootstrap=B) (bounding.py:381-400, 688-703): the replicas' streams come from four
            # words of the run's generator, drawn when the rebuild is decided (ns_prepare)
            bent = np.array([[rg.next64() for _ in range(4)]], dtype=np.uint64)
            f = float(ctx.bootstrap_expand(live_u[None], bent, bootstrap, bound == "multi")[0])
            if f > 1.0:
                bnd.scale_to_logvol(bnd.logvol + D * math.log(f))
        if enlarge != 1.0:
            bnd.scale_to_logvol(bnd.logvol + math.log(enlarge))
        nbound += 1

    def frames_of():
        if bound == "multi":
            return np.array(bnd.axes_ells, dtype=np.float64).copy(), np.array(bnd.logvol_ells, dtype=np.float64).copy()
        return np.array(bnd.axes, dtype=np.float64)[None].copy(), np.zeros(1)

    def cum_of(lv):  # ns_select: cumsum(exp(lv - max)) / total
        e = np.exp(lv - lv.max())
        c, out = 0.0, np.empty(len(lv))
        for j, x in enumerate(e):
            c += x
            out[j] = c
        return out * (1.0 / c)

    def pick(cum, xr):  # first index with cum >= xr, capped
        lo, hi = 0, len(cum) - 1
        while lo < hi:
            mid = (lo + hi) >> 1
            if cum[mid] < xr:
                lo = mid + 1
            else:
                hi = mid
        return lo

    fill = 0
    done = False
    while not done and fill < max_fills:
        it, ncall = int(state[0, 5]), int(state[0, 6])
        # ---- ns_prepare (sampler.py:625-674) ----
        eff = 100.0 * max(it, 1) / ncall
        want = (ncall >= first_ncall and eff < first_eff) if cube else (ncall >= ncall_last + update_interval or force)
        force = False
        if want:
            cube = False
            if forced == "exact" and undo is not None:
                # update_bound_if_needed runs inside _new_point when the queue's LAST entry has been popped
                # (sampler.py:771-772), before that entry's point replaces the worst one (sampler.py:1176-1185): the
                # bound is built from the live set without the newest point
                keep = live_u[undo[0]].copy()
                live_u[undo[0]] = undo[1]
                rebuild()
                live_u[undo[0]] = keep
            else:
                rebuild()
            ncall_last = ncall
            ev["rebuild_fills"].append(fill)
        # ---- ns_select: four words of the run's generator seed this fill's K selection streams ----
        ent = [rg.next64() for _ in range(4)]
        states = np.empty((K, 4), dtype=np.uint64)
        start = np.zeros(K, dtype=np.int64)
        xr = np.zeros(K)
        multi = (not cube) and bound == "multi" and bnd.nells > 1
        for w in range(K):
            g = Pcg()
            g.seed((ent[0] << 64) | ((ent[1] + w) & M64), (ent[2] << 64) | ((ent[3] + 2 * w) & M64))
            if not cube and sample != "unif":  # (the uniform sampler starts nowhere: internal_samplers.py:214-242)
                while True:
                    i = g.interval(N - 1)
                    if live_logl[i] > loglstar:
                        break
                start[w] = i
                if multi:
                    xr[w] = g.next_double()
            states[w] = g.words()
        if cube:
            out = ctx.unif_batch(prob, loglstar, states)
            q_nc = out["ncalls"].astype(np.int32)
        elif sample == "unif":
            if bound == "multi":
                out = ctx.unif_batch(prob, loglstar, states, ctrs=bnd.ctrs, axes=bnd.axes_ells, ams=bnd.ams,
                                     logvol_ells=bnd.logvol_ells)
            else:
                out = ctx.unif_batch(prob, loglstar, states, ctrs=bnd.ctr, axes=bnd.axes)
            q_nc = out["ncalls"].astype(np.int32)
            ta = tr = 0
        else:
            axes, lv = frames_of()
            cum = cum_of(lv) if multi else None
            fidx = np.array([pick(cum, xr[w]) if multi else 0 for w in range(K)], dtype=np.int32)
            u0 = live_u[start]
            inside = np.asarray(bnd.contains_many(u0)) if bound == "multi" else np.array([bnd.contains(x) for x in u0])
            if not inside.all():
                ev["forced_fills"].append(fill)
                if forced == "late":
                    force = True  # the run rebuilds before its NEXT fill
                else:
                    # sampler.py:484-489 inside the fill: entries up to the first one outside keep the old frames
                    jstar = int(np.argmin(inside))
                    rebuild()
                    ncall_last = ncall - carry  # self.ncall at the refill (sampler.py:631-632, 674)
                    new_axes, new_lv = frames_of()
                    multi2 = bound == "multi" and bnd.nells > 1
                    cum2 = cum_of(new_lv) if multi2 else None
                    nold = len(axes)
                    for w in range(jstar + 1, K):
                        if multi2:
                            # the entry's selection stream again: its start point, then the variate of its frame
                            g = Pcg()
                            g.seed((ent[0] << 64) | ((ent[1] + w) & M64), (ent[2] << 64) | ((ent[3] + 2 * w) & M64))
                            while True:
                                i = g.interval(N - 1)
                                if live_logl[i] > loglstar:
                                    break
                            fidx[w] = nold + pick(cum2, g.next_double())
                        else:
                            fidx[w] = nold
                    axes = np.concatenate([axes, new_axes])
                    assert np.asarray(bnd.contains_many(u0) if bound == "multi" else [bnd.contains(x) for x in u0]).all()
            if sample == "rwalk":
                out = ctx.rwalk_batch(prob, u0, axes, scale, loglstar, walks, states, axes_idx=fidx, bc=bc)
                q_nc = np.full(K, walks, dtype=np.int32)
                ta, tr = int(out["accept"].sum()), int(out["reject"].sum())
                # no step accepted = the start point again: its own stored ln L (the reference re-evaluates to the same
                # bits, internal_samplers.py:970-975; two device kernels may differ in the last one)
                out["logl"] = np.where(out["accept"] == 0, live_logl[start], out["logl"])
            else:
                out = ctx.slice_batch(prob, u0, axes, scale, loglstar, walks, states, principal=sample == "slice",
                                      doubling=doubling, axes_idx=fidx)
                q_nc = out["ncalls"].astype(np.int32)
                ta, tr = int(out["n_expand"].sum()), int(out["n_contract"].sum())
                if out["expansion_warning_set"].any():  # slice_doubling from the next fill on (:1188-1206)
                    doubling = True
        q_logl = np.ascontiguousarray(out["logl"], dtype=np.float64)
        res = ctx.ns_consume(live_l2, q_logl[None], q_nc[None], state, dlogz, live_it=live_it2, plateau=plateau)
        slots, srcs = res["dead_slot"][0].astype(np.int64), res["dead_src"][0].astype(np.int64)
        ev["dead_logl"].extend(res["dead_logl"][0].tolist())
        ev["dead_slot"].extend(slots.tolist())
        ev["dead_src"].extend(srcs.tolist())
        ev["fill_of_death"].extend([fill] * len(slots))
        carry = int(q_nc[srcs[-1] + 1:].sum()) if len(slots) else carry + int(q_nc.sum())
        undo = None
        if len(slots) and srcs[-1] == K - 1:
            sl = int(slots[-1])
            prev = [e for e in range(len(slots) - 1) if slots[e] == sl]
            undo = (sl, (out["u"][srcs[prev[-1]]] if prev else live_u[sl]).copy())
        if len(slots):
            order = np.argsort(slots, kind="stable")
            same = slots[order][1:] == slots[order][:-1]
            last = np.ones(len(slots), dtype=bool)
            last[order[:-1][same]] = False
            live_u[slots[last]] = out["u"][srcs[last]]
            live_v[slots[last]] = out["v"][srcs[last]]
        if not cube and sample == "rwalk":
            if ta + tr > 0:  # RWalkSampler.tune, once per fill (internal_samplers.py:460-493)
                scale *= math.exp((ta / (ta + tr) - facc) / D / facc)
        elif not cube and sample != "unif":  # tune_slice (internal_samplers.py:1209-1239)
            ne = float(max(ta, 1))
            scale *= min(max(ne * 2.0 / (ne + tr), 0.5), 2.0)
        loglstar = float(state[0, 7])
        fill += 1
        if res["stopped"][0] or np.ptp(live_logl) == 0:
            done = True
    # what Results reports: compute_integrals over the record with the reference loop's volume steps, plateaus included
    from oracle import nested_ref as R
    dead, ids = np.array(ev["dead_logl"]), np.array(ev["dead_slot"], dtype=np.int64)
    lv = R.logvol_from_record(dead, ids, live_logl.copy(), N)
    _, logz, _, _ = R.compute_integrals(np.concatenate([dead, np.sort(live_logl)]), lv)
    ev.update(niter=int(state[0, 5]), ncall=int(state[0, 6]), nbound=nbound, nfills=fill, logz=float(logz[-1]),
              live_logl=live_logl.copy(), live_u=live_u.copy(), done=done)
    return ev
