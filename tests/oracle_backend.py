"""TEST-ONLY backend: the methods of ``dynesty_amd._lib.Context`` that the
plugin classes call, implemented over the NumPy oracle.  It lets the CPU test
suite drive the *host* logic (plugin plumbing, pickling, dynesty integration,
multi-process sharding) where no GPU exists.  It is injected explicitly with
``dynesty_amd.backend.set_backend`` by tests; the product never imports it.
"""
import numpy as np

from oracle import bounding_ref as B
from oracle import friends_ref as F
from oracle import proposals_ref as P


def _gen(words):
    bg = np.random.PCG64()
    st = bg.state
    w = [int(x) for x in words]
    st["state"]["state"] = (w[0] << 64) | w[1]
    st["state"]["inc"] = (w[2] << 64) | w[3]
    st["has_uint32"] = 0
    st["uinteger"] = 0
    bg.state = st
    return np.random.Generator(bg)


def _words(gen):
    s = gen.bit_generator.state["state"]
    m = (1 << 64) - 1
    return np.array([s["state"] >> 64, s["state"] & m, s["inc"] >> 64,
                     s["inc"] & m], dtype=np.uint64)


def _canon(axes):
    out = np.array(axes, dtype=np.float64)
    for k in range(out.shape[1]):
        i = np.argmax(np.abs(out[:, k]))
        if out[i, k] < 0:
            out[:, k] = -out[:, k]
    return out


def _ell(ctr, cov, am, axes, axlens, logvol):
    return B.Ell(np.asarray(ctr), np.asarray(cov), np.asarray(am),
                 np.asarray(axes), np.asarray(axlens), float(logvol))


class OracleBackend:
    """canon=True: eigenvector signs fixed inside the split tree as on the device (see
    oracle.bounding_ref.CANON_SIGNS), so that ellipsoid lists come out in the device's order."""
    name = "oracle (tests only)"

    def __init__(self, canon=False):
        self.canon = bool(canon)

    def _signs(self):
        be = self

        class _Ctx:
            def __enter__(self):
                self.old = B.CANON_SIGNS
                B.CANON_SIGNS = be.canon or self.old

            def __exit__(self, *exc):
                B.CANON_SIGNS = self.old
                return False
        return _Ctx()

    def seed_children(self, entropy, first, k):
        kids = np.random.SeedSequence(
            [int(x) for x in np.atleast_1d(entropy)]).spawn(first + k)[first:]
        out = np.empty((k, 4), dtype=np.uint64)
        for i, c in enumerate(kids):
            out[i] = _words(np.random.Generator(np.random.PCG64(c)))
        return out

    def ns_consume(self, live_logl, q_logl, q_ncalls, state, dlogz, live_it=None, plateau=None):
        """dh_ns_consume through the oracle's restatement of the reference loop (plateau steps included;
        `plateau` carries the mode between calls as on the device)."""
        from oracle import nested_ref as R
        out = dict(dead_logl=[], dead_slot=[], dead_src=[], dead_it=[], dead_nc=[], stopped=[])
        for r in range(live_logl.shape[0]):
            s = R.RunState(live_logl.shape[1])
            s.logvol, s.logz, s.h, s.logzvar, s.loglstar = (float(x) for x in state[r, :5])
            s.it, s.ncall = int(state[r, 5]), int(state[r, 6])
            it = None if live_it is None else live_it[r].astype(np.int64)
            if plateau is not None and plateau[r, 0] > 0:
                s.plateau_mode, s.plateau_counter, s.plateau_logdvol = True, int(plateau[r, 0]), float(plateau[r, 1])
            res = R.consume_queue(live_logl[r], np.asarray(q_logl)[r], np.asarray(q_ncalls)[r], s, dlogz,
                                  plateau=True, live_it=it)
            if plateau is not None:
                plateau[r] = [s.plateau_counter if s.plateau_mode else 0, s.plateau_logdvol]
            if live_it is not None:
                live_it[r] = it
            state[r, :7] = [s.logvol, s.logz, s.h, s.logzvar, s.loglstar, s.it, s.ncall]
            state[r, 7] = live_logl[r].min()
            for k in ("dead_logl", "dead_slot", "dead_src", "dead_it", "dead_nc"):
                out[k].append(res[k])
            out["stopped"].append(res["stopped"])
        out["stopped"] = np.array(out["stopped"], dtype=bool)
        return out

    def problem_eval(self, prob, u):
        u = np.asarray(u, dtype=np.float64).reshape(-1, prob.ndim)
        v = prob.prior_transform_many(u)
        return v, prob.loglikelihood_many(v)

    def contains(self, x, ctrs, ams, mode=0, want_mask=False, want_quad=False):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[None]
        ctrs = np.asarray(ctrs).reshape(-1, x.shape[1])
        ams = np.asarray(ams).reshape(len(ctrs), x.shape[1], x.shape[1])
        quad = np.array([B.multi_quadforms(p, ctrs, ams) for p in x])
        inside = quad < 1 if mode == 0 else np.sqrt(quad) <= 1.0
        count = inside.sum(axis=1).astype(np.int32)
        mask = None
        if want_mask:
            k, m = inside.shape
            nw = (k + 63) // 64
            bits = np.zeros((m, nw * 64), dtype=np.uint8)
            bits[:, :k] = inside.T
            mask = np.packbits(bits, axis=1, bitorder="little").view(np.uint64)
        return count, mask, (quad if want_quad else None)

    def rebuild(self, points, multi=True, max_ells=None, want_labels=False):
        pts = np.asarray(points, dtype=np.float64)
        with self._signs():
            ells = B.multi_update(pts).ells if multi else \
                [B.bounding_ellipsoid(pts)]
        return dict(nells=len(ells), ctrs=np.array([e.ctr for e in ells]),
                    covs=np.array([e.cov for e in ells]),
                    ams=np.array([e.am for e in ells]),
                    axes=np.array([_canon(e.axes) for e in ells]),
                    axlens=np.array([e.axlens for e in ells]),
                    logvol_ells=np.array([e.logvol for e in ells]),
                    labels=None, nnodes=0)

    def ell_from_cov(self, covs):
        covs = np.asarray(covs, dtype=np.float64)
        if covs.ndim == 2:
            covs = covs[None]
        d = covs.shape[1]
        ells = [B.make_ell(np.zeros(d), c) for c in covs]
        return (np.array([_canon(e.axes) for e in ells]),
                np.array([e.axlens for e in ells]),
                np.array([e.am for e in ells]),
                np.array([e.logvol for e in ells]))

    def scale_to_logvol(self, covs, ams, axes, axlens, logvols, targets):
        d = axlens.shape[1]
        for i in range(len(logvols)):
            e = _ell(np.zeros(d), covs[i], ams[i], axes[i], axlens[i],
                     logvols[i])
            B.scale_ell_to_logvol(e, float(targets[i]))
            covs[i], ams[i], axes[i], axlens[i] = e.cov, e.am, e.axes, e.axlens
            logvols[i] = e.logvol

    def bound_draw(self, state4, nsamp, ctrs, axes, ams=None, logvol_ells=None,
                   return_q=False):
        rng = _gen(state4)
        ctrs = np.asarray(ctrs, dtype=np.float64)
        if ctrs.ndim == 1:
            ctrs = ctrs[None]
        m, d = ctrs.shape
        axes = np.asarray(axes).reshape(m, d, d)
        xs = np.empty((nsamp, d))
        idxs = np.zeros(nsamp, dtype=np.int32)
        qs = np.ones(nsamp, dtype=np.int32)
        if m == 1:
            e = _ell(ctrs[0], np.eye(d), np.eye(d), axes[0], np.ones(d), 0.0)
            for s in range(nsamp):
                xs[s] = B.ell_sample(e, rng)
        else:
            ams = np.asarray(ams).reshape(m, d, d)
            ells = [_ell(ctrs[i], np.eye(d), ams[i], axes[i], np.ones(d),
                         logvol_ells[i]) for i in range(m)]
            mell = B.stack_ells(ells)
            for s in range(nsamp):
                r = B.multi_sample(mell, rng, return_q=return_q)
                xs[s], idxs[s] = r[0], r[1]
                if return_q:
                    qs[s] = r[2]
        return xs, idxs, qs, _words(rng)

    @staticmethod
    def _bcmasks(bc, ndim):
        if bc is None:
            return None, None, None
        bc = np.asarray(bc)
        per = np.nonzero(bc == 1)[0]
        ref = np.nonzero(bc == 2)[0]
        return (per if len(per) else None, ref if len(ref) else None, bc == 0)

    def rwalk_batch(self, prob, u0, axes, scale, loglstar, walks, rng_states,
                    axes_idx=None, ncdim=None, bc=None):
        u0 = np.asarray(u0, dtype=np.float64).reshape(-1, prob.ndim)
        k = u0.shape[0]
        nc = prob.ndim if ncdim is None else ncdim
        axes = np.asarray(axes).reshape(-1, nc, nc)
        per, ref, nonb = self._bcmasks(bc, prob.ndim)
        out = dict(u=np.empty_like(u0), v=np.empty_like(u0), logl=np.empty(k),
                   accept=np.empty(k, np.int32), reject=np.empty(k, np.int32),
                   rng_out=np.empty((k, 4), np.uint64))
        for i in range(k):
            rng = _gen(rng_states[i])
            fr = axes[0 if axes_idx is None else axes_idx[i]]
            r = P.rwalk(u0[i].copy(), loglstar, fr, scale, prob.prior_transform,
                        prob.loglikelihood, rng, walks, periodic=per,
                        reflective=ref, nonbounded=nonb)
            out["u"][i], out["v"][i], out["logl"][i] = r["u"], r["v"], r["logl"]
            out["accept"][i], out["reject"][i] = r["accept"], r["reject"]
            out["rng_out"][i] = _words(rng)
        return out

    def rwalk_propose(self, u0, axes, scale, rng_states, axes_idx=None,
                      ncdim=None, bc=None):
        u0 = np.asarray(u0, dtype=np.float64)
        k, nd = u0.shape
        nc = nd if ncdim is None else ncdim
        axes = np.asarray(axes).reshape(-1, nc, nc)
        per, ref, nonb = self._bcmasks(bc, nd)
        up = np.array(u0)
        inside = np.zeros(k, dtype=bool)
        out = np.empty((k, 4), np.uint64)
        for i in range(k):
            rng = _gen(rng_states[i])
            fr = axes[0 if axes_idx is None else axes_idx[i]]
            p, fail = P.propose_ball(u0[i], scale, fr, nc, rng, per, ref, nonb)
            if not fail:
                up[i] = p
                inside[i] = True
            out[i] = _words(rng)
        return up, inside, out

    def slice_batch(self, prob, u0, axes, scale, loglstar, slices, rng_states,
                    principal=False, doubling=False, axes_idx=None):
        u0 = np.asarray(u0, dtype=np.float64).reshape(-1, prob.ndim)
        k = u0.shape[0]
        axes = np.asarray(axes).reshape(-1, prob.ndim, prob.ndim)
        fn = P.pslice if principal else P.rslice
        out = dict(u=np.empty_like(u0), v=np.empty_like(u0), logl=np.empty(k),
                   ncalls=np.empty(k, np.int32), n_expand=np.empty(k, np.int32),
                   n_contract=np.empty(k, np.int32),
                   expansion_warning_set=np.zeros(k, bool),
                   rng_out=np.empty((k, 4), np.uint64))
        for i in range(k):
            rng = _gen(rng_states[i])
            fr = axes[0 if axes_idx is None else axes_idx[i]]
            r = fn(u0[i].copy(), loglstar, fr, scale, prob.prior_transform,
                   prob.loglikelihood, rng, slices, doubling=doubling)
            out["u"][i], out["v"][i], out["logl"][i] = r["u"], r["v"], r["logl"]
            out["ncalls"][i] = r["ncalls"]
            out["n_expand"][i], out["n_contract"][i] = r["n_expand"], \
                r["n_contract"]
            out["expansion_warning_set"][i] = r["expansion_warning_set"]
            out["rng_out"][i] = _words(rng)
        return out

    def unif_batch(self, prob, loglstar, rng_states, ctrs=None, axes=None,
                   ams=None, logvol_ells=None, ncdim=None, bc=None,
                   max_tries=0):
        rng_states = np.asarray(rng_states).reshape(-1, 4)
        k = rng_states.shape[0]
        nd = prob.ndim
        nc = nd if ncdim is None else ncdim
        nonb = None if bc is None else (np.asarray(bc) == 0)
        out = dict(u=np.empty((k, nd)), v=np.empty((k, nd)), logl=np.empty(k),
                   ncalls=np.empty(k, np.int32),
                   rng_out=np.empty((k, 4), np.uint64))
        draw = None
        if ctrs is not None:
            ctrs = np.asarray(ctrs).reshape(-1, nc)
            m = len(ctrs)
            axes = np.asarray(axes).reshape(m, nc, nc)
            if m == 1:
                e = _ell(ctrs[0], np.eye(nc), np.eye(nc), axes[0], np.ones(nc),
                         0.0)
                draw = P.unif_single(e)
            else:
                ams = np.asarray(ams).reshape(m, nc, nc)
                mell = B.stack_ells([
                    _ell(ctrs[i], np.eye(nc), ams[i], axes[i], np.ones(nc),
                         logvol_ells[i]) for i in range(m)])
                draw = P.unif_multi(mell)
        for i in range(k):
            rng = _gen(rng_states[i])
            if draw is None:
                r = P.unitcube(loglstar, prob.prior_transform,
                               prob.loglikelihood, rng, nd)
            else:
                r = P.unif_bound(loglstar, draw, prob.prior_transform,
                                 prob.loglikelihood, rng, nd, nc,
                                 nonbounded=nonb)
            out["u"][i], out["v"][i], out["logl"][i] = r["u"], r["v"], r["logl"]
            out["ncalls"][i] = r["ncalls"]
            out["rng_out"][i] = _words(rng)
        return out

    # ---- RadFriends / SupFriends ---------------------------------------------------
    @staticmethod
    def _friends(kind, ctrs, axes, axes_inv):
        ctrs = np.asarray(ctrs, dtype=np.float64)
        d = ctrs.shape[1]
        return F.Friends(kind, np.eye(d), np.asarray(axes_inv) @ np.asarray(axes_inv),
                         np.asarray(axes), np.asarray(axes_inv), 0.0, ctrs)

    def friends_update(self, points, kind, am_prev=None, in_masks=None):
        points = np.asarray(points, dtype=np.float64)
        d = points.shape[1]
        fr = F.friends_init(kind, d)
        if am_prev is not None:
            fr.am = np.asarray(am_prev)
        if in_masks is None or not len(in_masks):
            out, info = F.friends_update(fr, points, use_clustering=am_prev is not None)
        else:
            # the oracle resamples from seeds; here the masks are given: same steps by hand
            if am_prev is not None:
                cov, ncl = F.covariance_from_clusters(points, fr.am)
            else:
                cov, ncl = np.cov(points, rowvar=False), 1
            from scipy import linalg as sla
            from scipy import spatial
            am, axes = sla.pinvh(cov), sla.sqrtm(cov)
            axes_inv = sla.pinvh(axes)
            pt = points @ axes_inv
            p = 2 if kind == 'balls' else np.inf
            r = max(max(spatial.KDTree(pt[m]).query(pt[~m], k=1, eps=0, p=p)[0])
                    for m in np.asarray(in_masks, dtype=bool))
            am2 = am / r**2
            out = F.Friends(kind, cov * r**2, am2, axes * r, axes_inv / r,
                            F.shape_logvol(kind, d, am2), points)
            info = dict(nclusters=ncl, rmax=float(r))
        return dict(cov=np.real(out.cov), am=np.real(out.am), axes=np.real(out.axes),
                    axes_inv=np.real(out.axes_inv), logvol=float(out.logvol),
                    rmax=info["rmax"], nclusters=info["nclusters"])

    def friends_within(self, ctrs, kind, axes_inv, x, want_bits=False):
        ctrs = np.asarray(ctrs, dtype=np.float64)
        n, d = ctrs.shape
        xs = np.asarray(x, dtype=np.float64).reshape(-1, d)
        fr = self._friends(kind, ctrs, np.eye(d), axes_inv)
        counts = np.zeros(len(xs), dtype=np.int32)
        bits = np.zeros((len(xs), (n + 63) // 64), dtype=np.uint64) if want_bits else None
        for i, xi in enumerate(xs):
            idx = F.friends_within(fr, xi)
            counts[i] = len(idx)
            if want_bits:
                b = np.zeros(bits.shape[1] * 64, dtype=np.uint8)
                b[idx] = 1
                bits[i] = np.packbits(b, bitorder="little").view(np.uint64)
        return counts, bits

    def friends_draw(self, state6, nsamp, ctrs, kind, axes, axes_inv, return_q=False):
        gen = _gen(state6[:4])
        st = gen.bit_generator.state
        st["has_uint32"], st["uinteger"] = int(state6[4]), int(state6[5])
        gen.bit_generator.state = st
        fr = self._friends(kind, ctrs, axes, axes_inv)
        xs = np.empty((nsamp, fr.ndim))
        qs = np.ones(nsamp, dtype=np.int32)
        for i in range(nsamp):
            if return_q:
                xs[i], qs[i] = F.friends_sample(fr, gen, return_q=True)
            else:
                xs[i] = F.friends_sample(fr, gen)
        st = gen.bit_generator.state
        out = np.concatenate([_words(gen), np.array([st["has_uint32"], st["uinteger"]],
                                                    dtype=np.uint64)])
        return xs, qs, out

    def unif_friends_batch(self, prob, loglstar, rng_states, ctrs, kind, axes,
                           axes_inv, bc=None, max_tries=0):
        rng_states = np.asarray(rng_states).reshape(-1, 4)
        k, nd = rng_states.shape[0], prob.ndim
        nonb = None if bc is None else (np.asarray(bc) == 0)
        fr = self._friends(kind, ctrs, axes, axes_inv)
        out = dict(u=np.empty((k, nd)), v=np.empty((k, nd)), logl=np.empty(k),
                   ncalls=np.empty(k, np.int32), rng_out=np.empty((k, 4), np.uint64))
        for i in range(k):
            rng = _gen(rng_states[i])
            r = P.unif_bound(loglstar, lambda g: F.friends_sample(fr, g),
                             prob.prior_transform, prob.loglikelihood, rng, nd, nd,
                             nonbounded=nonb)
            out["u"][i], out["v"][i], out["logl"][i] = r["u"], r["v"], r["logl"]
            out["ncalls"][i] = r["ncalls"]
            out["rng_out"][i] = _words(rng)
        return out

    def unif_propose(self, ndim, rng_states, ctrs=None, axes=None, ams=None,
                     logvol_ells=None, ncdim=None, bc=None, friends=None):
        rng_states = np.asarray(rng_states).reshape(-1, 6 if friends is not None else 4)
        k = rng_states.shape[0]
        nc = ndim if ncdim is None else ncdim
        nonb = None if bc is None else (np.asarray(bc) == 0)
        if friends is not None:
            kind, fc, fax, fai = friends
            fr = self._friends(kind, fc, fax, fai)
            draw = lambda g: F.friends_sample(fr, g)
        else:
            ctrs = np.asarray(ctrs).reshape(-1, nc)
            m = len(ctrs)
            axes = np.asarray(axes).reshape(m, nc, nc)
            if m == 1:
                draw = P.unif_single(_ell(ctrs[0], np.eye(nc), np.eye(nc), axes[0], np.ones(nc), 0.0))
            else:
                ams = np.asarray(ams).reshape(m, nc, nc)
                draw = P.unif_multi(B.stack_ells([
                    _ell(ctrs[i], np.eye(nc), ams[i], axes[i], np.ones(nc), logvol_ells[i]) for i in range(m)]))
        u = np.empty((k, ndim))
        out = np.empty((k, rng_states.shape[1]), np.uint64)
        for i in range(k):
            rng = _gen(rng_states[i][:4])
            if friends is not None:
                st = rng.bit_generator.state
                st["has_uint32"], st["uinteger"] = int(rng_states[i][4]), int(rng_states[i][5])
                rng.bit_generator.state = st
            while True:
                x = draw(rng)
                if P.unitcheck(x, None if nonb is None else nonb[:nc]):
                    break
            if nc != ndim:
                x = np.concatenate((x, rng.uniform(size=(ndim - nc))))
            u[i] = x
            out[i, :4] = _words(rng)
            if friends is not None:
                st = rng.bit_generator.state
                out[i, 4], out[i, 5] = st["has_uint32"], st["uinteger"]
        return u, out

    def slice_feed(self, kind, ndim, states6, consumed=None, nlook=0, axes=None,
                   axes_idx=None, scale=1.0):
        from dynesty_amd import _lib
        st = np.array(states6, dtype=np.uint64).reshape(-1, 6)
        k = st.shape[0]
        out = None
        if kind == 'direction':
            axes = np.asarray(axes, dtype=np.float64).reshape(-1, ndim, ndim)
            out = np.empty((k, ndim))
        elif kind == 'shuffle':
            out = np.empty((k, ndim), dtype=np.int32)
        look = np.empty((k, int(nlook)))
        for i in range(k):
            bg = np.random.PCG64()
            _lib.set_pcg_state6(bg, st[i])
            gen = np.random.Generator(bg)
            if consumed is not None and consumed[i]:
                gen.random(int(consumed[i]))
            if kind == 'direction':
                drhat = gen.standard_normal(size=ndim)
                drhat /= np.linalg.norm(drhat)
                fr = axes[0 if axes_idx is None else axes_idx[i]]
                out[i] = np.dot(fr, drhat) * scale
            elif kind == 'shuffle':
                idxs = np.arange(ndim)
                gen.shuffle(idxs)
                out[i] = idxs
            st[i] = _lib.pcg_state6(bg)
            if nlook:
                look[i] = gen.random(int(nlook))
        return st, out, look
