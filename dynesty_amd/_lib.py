"""ctypes binding of libdynhip.so (C ABI: include/dynhip.h).

There is deliberately **no CPU fallback**: if the shared library is missing or
no HIP device is visible, importing the backend raises.  Build the library
with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C dynesty_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ENV = "DYNHIP_LIB"

DH_OK = 0
ERR_VALUE, ERR_CONTAIN, ERR_REGION, ERR_SLICE = -1, -2, -3, -4
ERR_HIP, ERR_ARG, ERR_QZERO, ERR_NOMEM = -5, -6, -7, -8

BC_HARD, BC_PERIODIC, BC_REFLECT = 0, 1, 2


class DynHipError(RuntimeError):
    """HIP runtime / argument failures of libdynhip (no reference analogue)."""


def lib_path():
    p = os.environ.get(LIB_ENV)
    if p:
        return p
    return os.path.join(_HERE, "libdynhip.so")


_lib = None

_vp, _i, _u32, _u64, _dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
_pd = C.POINTER(C.c_double)

# name -> (restype, argtypes); kept in step with include/dynhip.h (checked by
# tests/test_abi.py, which parses the header).
SIGNATURES = {
    "dh_version": (_i, []),
    "dh_device_count": (_i, []),
    "dh_create": (_vp, [_i]),
    "dh_destroy": (None, [_vp]),
    "dh_last_error": (C.c_char_p, [_vp]),
    "dh_sync": (_i, [_vp]),
    "dh_stream": (_vp, [_vp]),
    "dh_malloc": (_vp, [_vp, _u64]),
    "dh_free": (_i, [_vp, _vp]),
    "dh_memcpy_h2d": (_i, [_vp, _vp, _vp, _u64]),
    "dh_memcpy_d2h": (_i, [_vp, _vp, _vp, _u64]),
    "dh_memset": (_i, [_vp, _vp, _i, _u64]),
    "dh_event_create": (_vp, [_vp]),
    "dh_event_destroy": (_i, [_vp, _vp]),
    "dh_event_record": (_i, [_vp, _vp]),
    "dh_event_elapsed_ms": (_i, [_vp, _vp, _vp, _pd]),
    "dh_problem_create": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i]),
    "dh_problem_destroy": (_i, [_vp, _i]),
    "dh_problem_eval": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "dh_seed_children": (_i, [_vp, _vp, _i, _u32, _i, _vp]),
    "dh_rng_stream": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "dh_contains": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "dh_rebuild": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                        _vp, _vp, _vp]),
    "dh_rebuild_batch_dev": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_rebuild_ragged_dev": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_ell_from_cov": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dh_improve_covar_mat": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dh_scale_to_logvol": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_enlarge_batch_dev": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                  _dbl]),
    "dh_rwalk_propose": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _vp, _vp,
                              _vp, _vp, _vp]),
    "dh_rwalk_batch": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl,
                            _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_rwalk_batch_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl,
                                _dbl, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp]),
    "dh_rwalk_batch_philox": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl, _i, _vp, _u64,
                                   _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "dh_rwalk_batch_philox_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl, _i, _vp, _u64,
                                       _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "dh_slice_batch": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl,
                            _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp]),
    "dh_slice_batch_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl,
                                _dbl, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp]),
    "dh_unif_batch": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _dbl,
                           _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "dh_unif_batch_dev": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                               _dbl, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                               _vp, _vp]),
    "dh_ns_consume": (_i, [_vp, _i, _i, _i, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp, _vp, _vp, _vp]),
    "dh_ns_ensemble": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _dbl, _dbl,
                            C.c_int64, C.c_int64, _vp, _i, _u32, _vp, _vp,
                            _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "dh_bootstrap_expand": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "dh_friends_update": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp]),
    "dh_friends_within": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "dh_friends_draw": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp,
                             _vp, _vp]),
    "dh_unif_friends_batch": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _dbl,
                                   _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp]),
    "dh_bound_draw": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp,
                           _vp, _vp, _vp]),
    "dh_slice_feed": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _dbl, _vp, _vp, _i, _vp,
                           _vp, _vp]),
    "dh_set_rwalk_form": (_i, [_vp, _i]),
    "dh_ns_set_option": (_i, [_vp, _i, _dbl]),
    "dh_ns_set_boundary": (_i, [_vp, _i, _vp]),
    "dh_set_rwalk_items": (_i, [_vp, _i, C.c_longlong]),
    "dh_slice_batch_philox": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl, _i, _i, _u64, _u64, _u64,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_slice_batch_philox_dev": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _dbl, _dbl, _i, _i, _u64, _u64,
                                       _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dh_unif_batch_philox": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _dbl, _vp, _u64, _u64, _u64,
                                  C.c_int64, _vp, _vp, _vp, _vp]),
    "dh_unif_batch_philox_dev": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _dbl, _vp, _u64, _u64, _u64,
                                      C.c_int64, _vp, _vp, _vp, _vp, _vp]),
}


def load():
    """Load libdynhip.so (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"libdynhip.so not found at {path}: the HIP extension is required "
            "(no CPU fallback). Build it with `make -C dynesty_amd/csrc` or "
            "__graft_entry__.build().")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def entropy_words(entropy):
    """numpy's SeedSequence coercion of a sequence of ints to uint32 words
    (bit_generator.pyx _coerce_to_uint32_array)."""
    out = []
    for n in np.atleast_1d(entropy).tolist():
        n = int(n)
        if n < 0:
            raise ValueError("entropy must be non-negative")
        if n == 0:
            out.append(0)
        while n > 0:
            out.append(n & 0xFFFFFFFF)
            n >>= 32
    return np.array(out, dtype=np.uint32)


def pcg_state_words(bitgen):
    """(4,) uint64 {state_hi, state_lo, inc_hi, inc_lo} of a numpy PCG64."""
    st = bitgen.state
    if st["bit_generator"] != "PCG64":
        raise TypeError("only numpy PCG64 streams are supported")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m], dtype=np.uint64)


def pcg_state6(bitgen):
    """(6,) uint64: pcg_state_words + {has_uint32, uinteger} (the buffered half
    of numpy's 32-bit draws, which Generator.integers(n) consumes)."""
    st = bitgen.state
    return np.concatenate([pcg_state_words(bitgen),
                           np.array([st["has_uint32"], st["uinteger"]],
                                    dtype=np.uint64)])


def set_pcg_state6(bitgen, words):
    st = bitgen.state
    w = [int(x) for x in words]
    st["state"]["state"] = (w[0] << 64) | w[1]
    st["state"]["inc"] = (w[2] << 64) | w[3]
    st["has_uint32"] = w[4]
    st["uinteger"] = w[5]
    bitgen.state = st


def set_pcg_state_words(bitgen, words):
    """Write a device-advanced state back into a numpy PCG64 (keeps the
    buffered uint32 half exactly as numpy would: see SURVEY.md appendix A)."""
    st = bitgen.state
    w = [int(x) for x in words]
    st["state"]["state"] = (w[0] << 64) | w[1]
    st["state"]["inc"] = (w[2] << 64) | w[3]
    bitgen.state = st


def enlarge_bootstrap_defaults(sample, enlarge, bootstrap):
    """(enlarge, bootstrap) of a run as the reference's _get_enlarge_bootstrap
    decides them (dynesty.py:169-200)."""
    if enlarge is not None and bootstrap is None:
        if enlarge < 1:
            raise ValueError("enlarge must be >= 1")
        return float(enlarge), 0
    if enlarge is None and bootstrap is not None:
        if not (bootstrap > 1 or bootstrap == 0):
            raise ValueError("bootstrap must be 0 or > 1")
        return 1.0, int(bootstrap)
    if enlarge is None and bootstrap is None:
        return (1.0, 5) if sample == 'unif' else (1.25, 0)
    if bootstrap == 0 or enlarge == 1:
        return float(enlarge), int(bootstrap)
    raise ValueError('Enlarge and bootstrap together do not make sense unless bootstrap=0 or enlarge = 1')


class Context:
    """One device + stream (dh_ctx).  Methods take/return NumPy arrays."""

    def __init__(self, device=0):
        self.lib = load()
        self.handle = self.lib.dh_create(int(device))
        if not self.handle:
            msg = self.lib.dh_last_error(None).decode()
            raise DynHipError(f"dh_create({device}) failed: {msg}")
        self.device = int(device)
        self._problems = {}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- errors -------------------------------------------------------------
    def _check(self, rc):
        if rc >= 0:
            return rc
        msg = self.lib.dh_last_error(self.handle).decode()
        if rc == ERR_VALUE:
            raise ValueError(msg)
        if rc in (ERR_CONTAIN, ERR_REGION, ERR_SLICE, ERR_QZERO):
            raise RuntimeError(msg)
        raise DynHipError(f"libdynhip error {rc}: {msg}")

    def set_rwalk_form(self, form):
        """0 (default) / 2: four lanes per walker + matrix cores (csrc/walkq.hip) wherever that kernel is
        built -- decided by the problem alone, never by the launch size; 1: one walker per lane always
        (csrc/walk.hip)."""
        self._check(self.lib.dh_set_rwalk_form(self.handle, int(form)))

    def set_rwalk_items(self, on=True, budget_bytes=0):
        """Four-lane rwalk: PCG64 item streams from a generator pass ahead of the walk (default) or drawn inside
        the walk kernel; `budget_bytes` > 0 bounds the pass's buffer (larger launches go in chunks of walkers)."""
        self._check(self.lib.dh_set_rwalk_items(self.handle, int(bool(on)), int(budget_bytes)))

    def sync(self):
        self._check(self.lib.dh_sync(self.handle))

    # -- device memory --------------------------------------------------------
    def malloc(self, nbytes):
        p = self.lib.dh_malloc(self.handle, int(nbytes))
        if not p:
            self._check(ERR_NOMEM)
        return p

    def free(self, dptr):
        self._check(self.lib.dh_free(self.handle, dptr))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self._check(self.lib.dh_memcpy_h2d(self.handle, p, _ptr(arr),
                                           arr.nbytes))
        return p

    def from_device(self, dptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.dh_memcpy_d2h(self.handle, _ptr(out), dptr,
                                           out.nbytes))
        return out

    def event(self):
        return self.lib.dh_event_create(self.handle)

    def record(self, ev):
        self._check(self.lib.dh_event_record(self.handle, ev))

    def elapsed_ms(self, ev0, ev1):
        ms = C.c_double(0.0)
        self._check(self.lib.dh_event_elapsed_ms(self.handle, ev0, ev1,
                                                 C.byref(ms)))
        return ms.value

    # -- problems -------------------------------------------------------------
    def problem(self, prob):
        """Upload (once) a dynesty_amd.problems.Problem; returns the handle."""
        key = id(prob)
        if key in self._problems:
            return self._problems[key][0]
        like_id, like_par, prior_id, prior_par = prob.device_spec()
        lp, pp = _f64(like_par), _f64(prior_par)
        h = self._check(self.lib.dh_problem_create(
            self.handle, prob.ndim, like_id, _ptr(lp), lp.size, prior_id,
            _ptr(pp) if pp.size else None, pp.size))
        self._problems[key] = (h, prob)  # keep prob alive: id() stays unique
        return h

    def problem_eval(self, prob, u):
        u = _f64(u).reshape(-1, prob.ndim)
        k = u.shape[0]
        v = np.empty_like(u)
        logl = np.empty(k)
        self._check(self.lib.dh_problem_eval(self.handle, self.problem(prob),
                                             k, _ptr(u), _ptr(v), _ptr(logl)))
        return v, logl

    # -- RNG ------------------------------------------------------------------
    def seed_children(self, entropy, first, k):
        words = entropy_words(entropy)
        out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_seed_children(self.handle, _ptr(words),
                                              words.size, int(first), int(k),
                                              _ptr(out)))
        return out

    def rng_stream(self, state4, n_normal, n_unif):
        st = np.ascontiguousarray(state4, dtype=np.uint64)
        nrm = np.empty(max(n_normal, 1))
        unf = np.empty(max(n_unif, 1))
        out = np.empty(4, dtype=np.uint64)
        self._check(self.lib.dh_rng_stream(self.handle, _ptr(st), n_normal,
                                           n_unif, _ptr(nrm), _ptr(unf),
                                           _ptr(out)))
        return nrm[:n_normal], unf[:n_unif], out

    # -- membership -------------------------------------------------------------
    def contains(self, x, ctrs, ams, mode=0, want_mask=False, want_quad=False):
        x = _f64(x)
        if x.ndim == 1:
            x = x[None, :]
        k, d = x.shape
        ctrs = _f64(ctrs).reshape(-1, d)
        m = ctrs.shape[0]
        ams = _f64(ams).reshape(m, d, d)
        count = np.empty(k, dtype=np.int32)
        nw = (k + 63) // 64
        mask = np.empty((m, nw), dtype=np.uint64) if want_mask else None
        quad = np.empty((k, m)) if want_quad else None
        self._check(self.lib.dh_contains(self.handle, _ptr(x), k, d,
                                         _ptr(ctrs), _ptr(ams), m, mode,
                                         _ptr(count), _ptr(mask), _ptr(quad)))
        return count, mask, quad

    # -- rebuild ------------------------------------------------------------------
    def rebuild(self, points, multi=True, max_ells=None, want_labels=False):
        """MultiEllipsoid.update / Ellipsoid.update (see dh_rebuild).  Returns a
        dict of stacked arrays for the resulting ellipsoids."""
        pts = _f64(points)
        n, d = pts.shape
        if max_ells is None:
            max_ells = max(1, n // (2 * d)) if multi else 1
        nells = C.c_int32(0)
        nnodes = C.c_int32(0)
        ctrs = np.empty((max_ells, d))
        covs = np.empty((max_ells, d, d))
        ams = np.empty((max_ells, d, d))
        axes = np.empty((max_ells, d, d))
        axlens = np.empty((max_ells, d))
        logvols = np.empty(max_ells)
        lop = np.empty(n, dtype=np.int32) if want_labels else None
        self._check(self.lib.dh_rebuild(
            self.handle, _ptr(pts), n, d, 0 if multi else 1, max_ells,
            C.byref(nells), _ptr(ctrs), _ptr(covs), _ptr(ams), _ptr(axes),
            _ptr(axlens), _ptr(logvols), _ptr(lop), C.byref(nnodes)))
        m = nells.value
        return dict(nells=m, ctrs=ctrs[:m].copy(), covs=covs[:m].copy(),
                    ams=ams[:m].copy(), axes=axes[:m].copy(),
                    axlens=axlens[:m].copy(), logvol_ells=logvols[:m].copy(),
                    labels=lop, nnodes=nnodes.value)

    def rebuild_many(self, point_sets, multi=True):
        """One launch for several point sets of different sizes (bootstrap
        replicas): returns a list of rebuild() dicts."""
        sets = [_f64(p) for p in point_sets]
        B = len(sets)
        d = sets[0].shape[1]
        if d > 44:  # wide-D path: one rebuild per set (the ragged batch is a narrow-D kernel)
            return [self.rebuild(p, multi=multi) for p in sets]
        n_arr = np.array([len(p) for p in sets], dtype=np.int32)
        n_max = int(n_arr.max())
        pad = np.zeros((B, n_max, d))
        for i, p in enumerate(sets):
            pad[i, :len(p)] = p
        me = max(1, n_max // (2 * d)) if multi else 1
        d_pts = self.to_device(pad)
        d_n = self.to_device(n_arr)
        sizes = dict(nells=B * 4, status=B * 4, ctrs=B * me * d * 8,
                     covs=B * me * d * d * 8, ams=B * me * d * d * 8,
                     axes=B * me * d * d * 8, axl=B * me * d * 8, lv=B * me * 8)
        buf = {k: self.malloc(v) for k, v in sizes.items()}
        try:
            self._check(self.lib.dh_rebuild_ragged_dev(
                self.handle, B, d_pts, n_max, d_n, d, 0 if multi else 1, me,
                buf["nells"], buf["status"], buf["ctrs"], buf["covs"],
                buf["ams"], buf["axes"], buf["axl"], buf["lv"]))
            nells = self.from_device(buf["nells"], (B,), np.int32)
            status = self.from_device(buf["status"], (B,), np.int32)
            ctrs = self.from_device(buf["ctrs"], (B, me, d), np.float64)
            covs = self.from_device(buf["covs"], (B, me, d, d), np.float64)
            ams = self.from_device(buf["ams"], (B, me, d, d), np.float64)
            axes = self.from_device(buf["axes"], (B, me, d, d), np.float64)
            axl = self.from_device(buf["axl"], (B, me, d), np.float64)
            lv = self.from_device(buf["lv"], (B, me), np.float64)
        finally:
            for p in list(buf.values()) + [d_pts, d_n]:
                self.free(p)
        out = []
        for i in range(B):
            if status[i] != 0:
                self.lib.dh_last_error(self.handle)
                raise RuntimeError(f"bootstrap replica {i}: rebuild failed "
                                   f"with code {int(status[i])}")
            m = int(nells[i])
            out.append(dict(nells=m, ctrs=ctrs[i, :m], covs=covs[i, :m],
                            ams=ams[i, :m], axes=axes[i, :m],
                            axlens=axl[i, :m], logvol_ells=lv[i, :m]))
        return out

    def ell_from_cov(self, covs):
        """Ellipsoid.__init__ numerics for a stack of covariances."""
        covs = _f64(covs)
        if covs.ndim == 2:
            covs = covs[None]
        m, d, _ = covs.shape
        axes = np.empty((m, d, d))
        axlens = np.empty((m, d))
        ams = np.empty((m, d, d))
        lvs = np.empty(m)
        self._check(self.lib.dh_ell_from_cov(self.handle, m, d, _ptr(covs),
                                             _ptr(axes), _ptr(axlens),
                                             _ptr(ams), _ptr(lvs)))
        return axes, axlens, ams, lvs

    def improve_covar_mat(self, covs):
        """improve_covar_mat (bounding.py:1311-1384) for a stack of matrices:
        returns (good (m,) bool, cov, am, axes)."""
        covs = _f64(covs)
        if covs.ndim == 2:
            covs = covs[None]
        m, d, _ = covs.shape
        good = np.empty(m, dtype=np.int32)
        out = np.empty((m, d, d)); am = np.empty((m, d, d)); axes = np.empty((m, d, d))
        self._check(self.lib.dh_improve_covar_mat(self.handle, m, d, _ptr(covs), _ptr(good),
                                                  _ptr(out), _ptr(am), _ptr(axes)))
        return good.astype(bool), out, am, axes

    def scale_to_logvol(self, covs, ams, axes, axlens, logvols, targets):
        """In-place Ellipsoid.scale_to_logvol for a stack of ellipsoids; the
        arrays must be C-contiguous float64 (they are updated in place)."""
        m, d = axlens.shape
        for a in (covs, ams, axes, axlens, logvols):
            assert a.flags.c_contiguous and a.dtype == np.float64
        t = _f64(targets).reshape(m)
        self._check(self.lib.dh_scale_to_logvol(
            self.handle, m, d, _ptr(covs), _ptr(ams), _ptr(axes), _ptr(axlens),
            _ptr(logvols), _ptr(t)))

    # -- proposals ----------------------------------------------------------------
    def rwalk_batch(self, prob, u0, axes, scale, loglstar, walks, rng_states,
                    axes_idx=None, ncdim=None, bc=None):
        """Batched RWalkSampler.sample (see dh_rwalk_batch)."""
        ndim = prob.ndim
        u0 = _f64(u0).reshape(-1, ndim)
        k = u0.shape[0]
        ncdim = ndim if ncdim is None else int(ncdim)
        axes = _f64(axes).reshape(-1, ncdim, ncdim)
        m = axes.shape[0]
        idx = None if axes_idx is None else np.ascontiguousarray(
            axes_idx, dtype=np.int32)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        rng = np.ascontiguousarray(rng_states, dtype=np.uint64).reshape(k, 4)
        u = np.empty((k, ndim))
        v = np.empty((k, ndim))
        logl = np.empty(k)
        nacc = np.empty(k, dtype=np.int32)
        nrej = np.empty(k, dtype=np.int32)
        rng_out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_rwalk_batch(
            self.handle, self.problem(prob), k, ndim, ncdim, _ptr(u0),
            _ptr(axes), m, _ptr(idx), float(scale), float(loglstar),
            int(walks), _ptr(bcarr), _ptr(rng), _ptr(u), _ptr(v), _ptr(logl),
            _ptr(nacc), _ptr(nrej), _ptr(rng_out)))
        return dict(u=u, v=v, logl=logl, accept=nacc, reject=nrej,
                    rng_out=rng_out)


    def rwalk_batch_philox(self, prob, u0, axes, scale, loglstar, walks, seed,
                           sequence0=0, offset=0, axes_idx=None, ncdim=None, bc=None):
        """Throughput mode of the batched RWalkSampler.sample (dh_rwalk_batch_philox):
        hiprand Philox4x32-10 keyed (seed, sequence0 + walker, offset)."""
        ndim = prob.ndim
        u0 = _f64(u0).reshape(-1, ndim)
        k = u0.shape[0]
        ncdim = ndim if ncdim is None else int(ncdim)
        axes = _f64(axes).reshape(-1, ncdim, ncdim)
        idx = None if axes_idx is None else np.ascontiguousarray(axes_idx, dtype=np.int32)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        u = np.empty((k, ndim)); v = np.empty((k, ndim)); logl = np.empty(k)
        nacc = np.empty(k, dtype=np.int32); nrej = np.empty(k, dtype=np.int32)
        self._check(self.lib.dh_rwalk_batch_philox(
            self.handle, self.problem(prob), k, ndim, ncdim, _ptr(u0), _ptr(axes), axes.shape[0],
            _ptr(idx), float(scale), float(loglstar), int(walks), _ptr(bcarr), int(seed),
            int(sequence0), int(offset), _ptr(u), _ptr(v), _ptr(logl), _ptr(nacc), _ptr(nrej)))
        return dict(u=u, v=v, logl=logl, accept=nacc, reject=nrej)

    def rwalk_propose(self, u0, axes, scale, rng_states, axes_idx=None,
                      ncdim=None, bc=None):
        """One propose_ball_point per walker (dh_rwalk_propose)."""
        u0 = _f64(u0)
        k, ndim = u0.shape
        ncdim = ndim if ncdim is None else int(ncdim)
        axes = _f64(axes).reshape(-1, ncdim, ncdim)
        idx = None if axes_idx is None else np.ascontiguousarray(
            axes_idx, dtype=np.int32)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        rng = np.ascontiguousarray(rng_states, dtype=np.uint64).reshape(k, 4)
        up = np.empty((k, ndim))
        inside = np.empty(k, dtype=np.int32)
        rng_out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_rwalk_propose(
            self.handle, k, ndim, ncdim, _ptr(u0), _ptr(axes), axes.shape[0],
            _ptr(idx), float(scale), _ptr(bcarr), _ptr(rng), _ptr(up),
            _ptr(inside), _ptr(rng_out)))
        return up, inside.astype(bool), rng_out

    def slice_feed(self, kind, ndim, states6, consumed=None, nlook=0, axes=None,
                   axes_idx=None, scale=1.0):
        """Direction / axis order + uncommitted uniform lookahead of every
        walker's stream for one slice (dh_slice_feed).  kind: 'direction'
        (rslice), 'shuffle' (slice) or 'advance'.  Returns
        (states6, dirs | perm | None, look)."""
        kd = {'direction': 0, 'shuffle': 1, 'advance': 2}[kind]
        st = np.array(states6, dtype=np.uint64).reshape(-1, 6)
        k = st.shape[0]
        cons = None if consumed is None else np.ascontiguousarray(
            consumed, dtype=np.int32)
        ax = idx = dirs = perm = None
        m = 0
        if kd == 0:
            ax = _f64(axes).reshape(-1, ndim, ndim)
            m = ax.shape[0]
            idx = None if axes_idx is None else np.ascontiguousarray(
                axes_idx, dtype=np.int32)
            dirs = np.empty((k, ndim))
        elif kd == 1:
            perm = np.empty((k, ndim), dtype=np.int32)
        look = np.empty((k, int(nlook)))
        self._check(self.lib.dh_slice_feed(
            self.handle, k, int(ndim), kd, _ptr(ax), m, _ptr(idx), float(scale),
            _ptr(st), _ptr(cons), int(nlook), _ptr(dirs), _ptr(perm),
            _ptr(look) if nlook else None))
        return st, (dirs if kd == 0 else perm), look

    def slice_batch(self, prob, u0, axes, scale, loglstar, slices, rng_states,
                    principal=False, doubling=False, axes_idx=None):
        """Batched RSliceSampler.sample / SliceSampler.sample (dh_slice_batch)."""
        ndim = prob.ndim
        u0 = _f64(u0).reshape(-1, ndim)
        k = u0.shape[0]
        axes = _f64(axes).reshape(-1, ndim, ndim)
        idx = None if axes_idx is None else np.ascontiguousarray(
            axes_idx, dtype=np.int32)
        rng = np.ascontiguousarray(rng_states, dtype=np.uint64).reshape(k, 4)
        u = np.empty((k, ndim))
        v = np.empty((k, ndim))
        logl = np.empty(k)
        nc = np.empty(k, dtype=np.int32)
        ne = np.empty(k, dtype=np.int32)
        nt = np.empty(k, dtype=np.int32)
        fl = np.empty(k, dtype=np.int32)
        rng_out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_slice_batch(
            self.handle, self.problem(prob), k, ndim, 1 if principal else 0,
            _ptr(u0), _ptr(axes), axes.shape[0], _ptr(idx), float(scale),
            float(loglstar), int(slices), 1 if doubling else 0, _ptr(rng),
            _ptr(u), _ptr(v), _ptr(logl), _ptr(nc), _ptr(ne), _ptr(nt),
            _ptr(fl), _ptr(rng_out)))
        return dict(u=u, v=v, logl=logl, ncalls=nc, n_expand=ne, n_contract=nt,
                    expansion_warning_set=(fl & 1).astype(bool),
                    rng_out=rng_out)

    def slice_batch_philox(self, prob, u0, axes, scale, loglstar, slices, seed, sequence0=0, offset=0,
                           axes_idx=None, principal=False, doubling=False):
        """Throughput mode of the batched slice samplers (dh_slice_batch_philox): hiprand
        Philox4x32-10 keyed (seed, sequence0 + walker, offset)."""
        ndim = prob.ndim
        u0 = _f64(u0).reshape(-1, ndim)
        k = u0.shape[0]
        axes = _f64(axes).reshape(-1, ndim, ndim)
        idx = None if axes_idx is None else np.ascontiguousarray(axes_idx, dtype=np.int32)
        u = np.empty((k, ndim)); v = np.empty((k, ndim)); logl = np.empty(k)
        nc = np.empty(k, dtype=np.int32); ne = np.empty(k, dtype=np.int32)
        nt = np.empty(k, dtype=np.int32); fl = np.empty(k, dtype=np.int32)
        self._check(self.lib.dh_slice_batch_philox(
            self.handle, self.problem(prob), k, ndim, 1 if principal else 0, _ptr(u0), _ptr(axes),
            axes.shape[0], _ptr(idx), float(scale), float(loglstar), int(slices), 1 if doubling else 0,
            int(seed), int(sequence0), int(offset), _ptr(u), _ptr(v), _ptr(logl), _ptr(nc), _ptr(ne),
            _ptr(nt), _ptr(fl)))
        return dict(u=u, v=v, logl=logl, ncalls=nc, n_expand=ne, n_contract=nt,
                    expansion_warning_set=(fl & 1).astype(bool))

    def unif_batch_philox(self, prob, loglstar, k, seed, sequence0=0, offset=0, ctrs=None, axes=None,
                          ams=None, logvol_ells=None, ncdim=None, bc=None, max_tries=0):
        """Throughput mode of the batched UniformBoundSampler.sample / UnitCubeSampler.sample
        (dh_unif_batch_philox) for k walkers."""
        ndim = prob.ndim
        ncdim = ndim if ncdim is None else int(ncdim)
        if ctrs is None:
            m, c, ax, am, cp = 0, None, None, None, None
        else:
            c = _f64(ctrs).reshape(-1, ncdim)
            m = c.shape[0]
            ax = _f64(axes).reshape(m, ncdim, ncdim)
            am = cp = None
            if m > 1:
                am = _f64(ams).reshape(m, ncdim, ncdim)
                cp = cumprob_of(logvol_ells)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        u = np.empty((k, ndim)); v = np.empty((k, ndim)); logl = np.empty(k)
        nc = np.empty(k, dtype=np.int32)
        self._check(self.lib.dh_unif_batch_philox(
            self.handle, self.problem(prob), int(k), ndim, ncdim, m, _ptr(c), _ptr(ax), _ptr(am), _ptr(cp),
            float(loglstar), _ptr(bcarr), int(seed), int(sequence0), int(offset), int(max_tries),
            _ptr(u), _ptr(v), _ptr(logl), _ptr(nc)))
        return dict(u=u, v=v, logl=logl, ncalls=nc)

    def unif_batch(self, prob, loglstar, rng_states, ctrs=None, axes=None,
                   ams=None, logvol_ells=None, ncdim=None, bc=None,
                   max_tries=0):
        """Batched UniformBoundSampler.sample (ellipsoid bound given by
        ctrs/axes[/ams/logvol_ells]) or UnitCubeSampler.sample (ctrs=None)."""
        ndim = prob.ndim
        rng = np.ascontiguousarray(rng_states, dtype=np.uint64).reshape(-1, 4)
        k = rng.shape[0]
        ncdim = ndim if ncdim is None else int(ncdim)
        if ctrs is None:
            m, c, ax, am, cp = 0, None, None, None, None
        else:
            c = _f64(ctrs).reshape(-1, ncdim)
            m = c.shape[0]
            ax = _f64(axes).reshape(m, ncdim, ncdim)
            am = cp = None
            if m > 1:
                am = _f64(ams).reshape(m, ncdim, ncdim)
                cp = cumprob_of(logvol_ells)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        u = np.empty((k, ndim))
        v = np.empty((k, ndim))
        logl = np.empty(k)
        nc = np.empty(k, dtype=np.int32)
        rng_out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_unif_batch(
            self.handle, self.problem(prob), k, ndim, ncdim, m, _ptr(c),
            _ptr(ax), _ptr(am), _ptr(cp), float(loglstar), _ptr(bcarr),
            _ptr(rng), int(max_tries), _ptr(u), _ptr(v), _ptr(logl), _ptr(nc),
            _ptr(rng_out)))
        return dict(u=u, v=v, logl=logl, ncalls=nc, rng_out=rng_out)

    def unif_propose(self, ndim, rng_states, ctrs=None, axes=None, ams=None,
                     logvol_ells=None, ncdim=None, bc=None, friends=None):
        """Lock-step half of UniformBoundSampler for an arbitrary host
        likelihood: per walker the next candidate of its stream that lies in
        the bound and passes unitcheck (dh_unif_batch / dh_unif_friends_batch
        with problem = -1).  friends = (kind, ctrs, axes, axes_inv) selects a
        balls / cubes bound; rng_states then has 6 columns (the last two carry
        NumPy's buffered 32-bit half, which integers(n) consumes) and so has
        the returned state.  Returns (u (k, ndim), rng_out)."""
        rs = np.ascontiguousarray(rng_states, dtype=np.uint64)
        rs = rs.reshape(-1, 6 if friends is not None else 4)
        rng = np.ascontiguousarray(rs[:, :4])
        k = rng.shape[0]
        ncdim = ndim if ncdim is None else int(ncdim)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        u = np.empty((k, ndim))
        rng_out = np.empty((k, 4), dtype=np.uint64)
        if friends is not None:
            kind, fc, fax, fai = friends
            c = _f64(fc).reshape(-1, ndim)
            r32 = np.ascontiguousarray(rs[:, 4:])
            r32o = np.empty((k, 2), dtype=np.uint64)
            self._check(self.lib.dh_unif_friends_batch(
                self.handle, -1, k, ndim, 0 if kind == 'balls' else 1, _ptr(c),
                c.shape[0], _ptr(_f64(fax)), _ptr(_f64(fai)), 0.0, _ptr(bcarr),
                _ptr(rng), 0, _ptr(u), None, None, None, _ptr(rng_out),
                _ptr(r32), _ptr(r32o)))
            return u, np.concatenate([rng_out, r32o], axis=1)
        if ctrs is None:
            m, c, ax, am, cp = 0, None, None, None, None
        else:
            c = _f64(ctrs).reshape(-1, ncdim)
            m = c.shape[0]
            ax = _f64(axes).reshape(m, ncdim, ncdim)
            am = cp = None
            if m > 1:
                am = _f64(ams).reshape(m, ncdim, ncdim)
                cp = cumprob_of(logvol_ells)
        self._check(self.lib.dh_unif_batch(
            self.handle, -1, k, ndim, ncdim, m, _ptr(c), _ptr(ax), _ptr(am),
            _ptr(cp), 0.0, _ptr(bcarr), _ptr(rng), 0, _ptr(u), None, None,
            None, _ptr(rng_out)))
        return u, rng_out

    def unif_friends_batch(self, prob, loglstar, rng_states, ctrs, kind, axes,
                           axes_inv, bc=None, max_tries=0):
        """Batched UniformBoundSampler.sample inside a RadFriends ('balls') /
        SupFriends ('cubes') bound (dh_unif_friends_batch)."""
        ndim = prob.ndim
        rng = np.ascontiguousarray(rng_states, dtype=np.uint64).reshape(-1, 4)
        k = rng.shape[0]
        c = _f64(ctrs).reshape(-1, ndim)
        bcarr = None if bc is None else np.ascontiguousarray(bc, dtype=np.int8)
        u = np.empty((k, ndim))
        v = np.empty((k, ndim))
        logl = np.empty(k)
        nc = np.empty(k, dtype=np.int32)
        rng_out = np.empty((k, 4), dtype=np.uint64)
        self._check(self.lib.dh_unif_friends_batch(
            self.handle, self.problem(prob), k, ndim,
            0 if kind == 'balls' else 1, _ptr(c), c.shape[0],
            _ptr(_f64(axes)), _ptr(_f64(axes_inv)), float(loglstar),
            _ptr(bcarr), _ptr(rng), int(max_tries), _ptr(u), _ptr(v),
            _ptr(logl), _ptr(nc), _ptr(rng_out), None, None))
        return dict(u=u, v=v, logl=logl, ncalls=nc, rng_out=rng_out)

    def ns_consume(self, live_logl, q_logl, q_ncalls, state, dlogz, live_it=None, plateau=None):
        """One queue consumption per run (dh_ns_consume).  live_logl (R, N) and
        state (R, 8) are updated in place; returns dict(dead_logl, dead_slot,
        dead_src (lists per run), stopped (R,) bool).  With live_it ((R, N) int32,
        updated in place: iteration at which each live point was proposed) also
        dead_it / dead_nc per run (the reference's per-point 'it' and 'nc').
        plateau: optional (R, 2) float64, updated in place -- the reference's
        likelihood-plateau mode carried between calls (deaths left, ln of the
        plateau's volume step; sampler.py:1112-1127)."""
        live = np.ascontiguousarray(live_logl, dtype=np.float64)
        assert live is live_logl and live.ndim == 2
        R, N = live.shape
        ql = _f64(q_logl).reshape(R, -1)
        K = ql.shape[1]
        qn = np.ascontiguousarray(q_ncalls, dtype=np.int32).reshape(R, K)
        assert state.dtype == np.float64 and state.shape == (R, 8) and state.flags.c_contiguous
        dl = np.empty((R, K)); ds = np.empty((R, K), dtype=np.int32); dj = np.empty((R, K), dtype=np.int32)
        nd = np.empty(R, dtype=np.int32); stp = np.empty(R, dtype=np.int32)
        dit = dnc = None
        if live_it is not None:
            assert live_it.dtype == np.int32 and live_it.shape == (R, N) and live_it.flags.c_contiguous
            dit = np.empty((R, K), dtype=np.int32)
            dnc = np.empty((R, K), dtype=np.int32)
        if plateau is not None:
            assert plateau.dtype == np.float64 and plateau.shape == (R, 2) and plateau.flags.c_contiguous
        self._check(self.lib.dh_ns_consume(self.handle, R, N, K, float(dlogz), _ptr(live), _ptr(ql),
                                           _ptr(qn), _ptr(state), _ptr(dl), _ptr(ds), _ptr(dj),
                                           _ptr(nd), _ptr(stp), _ptr(live_it), _ptr(dit), _ptr(dnc),
                                           _ptr(plateau)))
        out = dict(dead_logl=[dl[r, :nd[r]].copy() for r in range(R)],
                   dead_slot=[ds[r, :nd[r]].copy() for r in range(R)],
                   dead_src=[dj[r, :nd[r]].copy() for r in range(R)],
                   stopped=stp.astype(bool))
        if live_it is not None:
            out["dead_it"] = [dit[r, :nd[r]].copy() for r in range(R)]
            out["dead_nc"] = [dnc[r, :nd[r]].copy() for r in range(R)]
        return out

    def ns_ensemble(self, prob, runs, nlive, queue_size, walks=None,
                    bound='multi', dlogz=0.01, enlarge=None, entropy=(21,),
                    first_run=0, max_fills=0, max_iter=400000,
                    want_dead_logl=False, sample='rwalk', slices=None,
                    rebuild_sync=False, want_samples=False, rng='pcg64', bootstrap=None, rebuild_every=0,
                    update_interval=None, first_update=None, maxiter=None, maxcall=None, logl_max=None,
                    add_live=True, forced_exact=True, periodic=None, reflective=None):
        """Device-resident ensemble of static NS runs (dh_ns_ensemble).

        periodic / reflective: lists of coordinate indices as NestedSampler takes them (dynesty.py:297-310); they reach
        the rwalk and uniform samplers (dh_ns_set_boundary), the slice samplers ignore them as the reference's do.

        forced_exact=True (the default): the reference's protocol -- propose_live's forced bound update
        (sampler.py:484-489) belongs to the fill that finds a start point outside the bound, and the regular bound is
        built before the newest point enters (DH_NS_OPT_FORCED_EXACT).  False: the late form (the run is flagged and
        rebuilds before its next fill): 9-20 % fewer bound updates than the reference, a few per cent faster.

        update_interval / first_update (NestedSampler, dynesty.py:213-234: a float update_interval is a multiple of
        nlive, an int a number of calls; first_update = dict(min_ncall=..., min_eff=...)) and maxiter / maxcall /
        logl_max / add_live (run_nested, sampler.py:1214-1300) as the reference takes them; None = its default.

        rebuild_every=n: bounds are built every n-th fill and runs that become due in
        between wait -- per-run results unchanged (bit-identical with PCG64 streams),
        fewer and fuller rebuild fills; 0 = chosen from the run's shape.

        sample: 'rwalk' | 'rslice' | 'slice' | 'unif'.  enlarge / bootstrap default as the reference's
        _get_enlarge_bootstrap (dynesty.py:169-200): (1.25, 0), and (1, 5) for 'unif'.

        rebuild_sync=False keeps the reference's per-run update schedule
        (sampler.py:625-674): a run's result then depends only on its own seed,
        not on how the ensemble is sharded.  rebuild_sync=True lets all runs
        that already have a bound rebuild it together whenever any run is due
        (early, never late): fewer, fuller rebuild launches, but a run's
        schedule then depends on its shard mates."""
        nd = prob.ndim
        if bound not in ('multi', 'single'):
            raise ValueError(f"ns_ensemble: bound={bound!r} is not supported by the device-resident "
                             "loop ('multi' or 'single'; balls / cubes: nested.run_static)")
        if sample not in ('rwalk', 'rslice', 'slice', 'unif'):
            raise ValueError(f"ns_ensemble: sample={sample!r} is not supported by the "
                             "device-resident loop ('rwalk', 'rslice', 'slice' or 'unif')")
        kind = dict(rwalk=0, rslice=1, slice=2, unif=6)[sample]
        if rng not in ('pcg64', 'philox'):
            raise ValueError("ns_ensemble: rng must be 'pcg64' or 'philox'")
        enlarge, bootstrap = enlarge_bootstrap_defaults(sample, enlarge, bootstrap)
        if kind == 6:
            walks = 1  # unused
        elif kind == 0:
            if walks is None:
                walks = nd + 20  # dynesty.py:128
        else:
            walks = slices if slices is not None else \
                (3 + nd if kind == 1 else 3)
        words = entropy_words(entropy)
        rec = np.empty((runs, 8))
        want_dead_logl = want_dead_logl or want_samples
        dead = np.empty((runs, max_iter)) if want_dead_logl else None
        livel = np.empty((runs, nlive)) if want_dead_logl else None
        # runs x max_iter x ndim: only the niter rows a run produced are touched
        dead_u = np.empty((runs, max_iter, nd)) if want_samples else None
        live_u = np.empty((runs, nlive, nd)) if want_samples else None
        # the reference's per-point id / it / nc (sampler.py:1165-1182) travel with the samples
        pid = np.empty((runs, max_iter), dtype=np.int32) if want_samples else None
        pit = np.empty((runs, max_iter), dtype=np.int32) if want_samples else None
        pnc = np.empty((runs, max_iter), dtype=np.int32) if want_samples else None
        lit = np.empty((runs, nlive), dtype=np.int32) if want_samples else None
        nf = C.c_int64(0)
        nan = float('nan')
        if update_interval is not None:
            # dynesty.py:213-234: a float is a multiple of nlive, an int a number of calls
            # (np.floating is not a Python float: an np.float32 ratio used to be truncated by int(); any value is
            # clamped to >= 1 call like the reference's max(min(round(ratio * nlive), maxsize), 1), dynesty.py:646-649)
            if isinstance(update_interval, (int, np.integer)) and not isinstance(update_interval, bool):
                update_interval = max(1, int(update_interval))
            else:
                update_interval = max(1, int(round(float(update_interval) * nlive)))
        fu = first_update or {}
        opts = [nan if update_interval is None else float(update_interval),
                float(fu['min_ncall']) if 'min_ncall' in fu else nan,
                float(fu['min_eff']) if 'min_eff' in fu else nan,
                nan if maxiter is None else float(maxiter), nan if maxcall is None else float(maxcall),
                nan if logl_max is None else float(logl_max), nan if add_live else 0.0,
                1.0 if forced_exact else 0.0]
        for key, val in enumerate(opts):
            self._check(self.lib.dh_ns_set_option(self.handle, key, val))
        bc = None
        if periodic is not None or reflective is not None:
            bc = np.zeros(nd, dtype=np.int8)
            if periodic is not None:
                bc[np.asarray(periodic, dtype=int)] = BC_PERIODIC
            if reflective is not None:
                bc[np.asarray(reflective, dtype=int)] = BC_REFLECT
        self._check(self.lib.dh_ns_set_boundary(self.handle, nd if bc is not None else 0, _ptr(bc)))
        self._check(self.lib.dh_ns_ensemble(
            self.handle, self.problem(prob), int(runs), int(nlive), nd,
            int(queue_size), kind + ((1 if kind == 6 else 3) if rng == 'philox' else 0), int(walks),
            1 if bound == 'multi' else 0,
            1 if rebuild_sync else 0, float(dlogz), float(enlarge), int(max_fills), int(max_iter),
            _ptr(words), words.size, int(first_run), _ptr(rec), _ptr(dead),
            _ptr(livel), _ptr(dead_u), _ptr(live_u), C.byref(nf), _ptr(pid), _ptr(pit), _ptr(pnc),
            _ptr(lit), int(bootstrap), int(rebuild_every)))
        out = dict(logz=rec[:, 0], logzerr=rec[:, 1],
                   niter=rec[:, 2].astype(np.int64),
                   ncall=rec[:, 3].astype(np.int64), h=rec[:, 4],
                   nbound=rec[:, 5].astype(np.int64),
                   status=rec[:, 6].astype(np.int64), eff=rec[:, 7],
                   nfills=nf.value)
        if want_dead_logl:
            out["dead_logl"] = dead
            out["live_logl"] = livel
        if want_samples:
            out["dead_u"] = dead_u
            out["live_u"] = live_u
            out["dead_id"], out["dead_it"], out["dead_nc"], out["live_it"] = pid, pit, pnc, lit
        return out

    # ---- RadFriends / SupFriends ------------------------------------------
    def bootstrap_expand(self, points, ent, bootstrap, multi, want_n_in=False):
        """dh_bootstrap_expand: points (runs, n, d), ent (runs, 4) uint64 ->
        (runs,) expansion factors (bounding.py:1619-1648, max over replicas)
        [, (runs, bootstrap) sample sizes]."""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        if pts.ndim == 2:
            pts = pts[None]
        runs, n, d = pts.shape
        ent = np.ascontiguousarray(ent, dtype=np.uint64).reshape(runs, 4)
        out = np.empty(runs)
        nin = np.empty((runs, int(bootstrap)), dtype=np.int32) if want_n_in else None
        self._check(self.lib.dh_bootstrap_expand(self.handle, runs, _ptr(pts), n, d, 1 if multi else 0,
                                                 int(bootstrap), _ptr(ent), _ptr(out), _ptr(nin)))
        return (out, nin) if want_n_in else out

    def friends_update(self, points, kind, am_prev=None, in_masks=None):
        """RadFriends.update / SupFriends.update (dh_friends_update).
        kind 'balls' | 'cubes'; am_prev = previous metric (None: no
        clustering); in_masks = (B, n) bool resampling masks or None (LOO)."""
        pts = _f64(points)
        n, d = pts.shape
        prev = None if am_prev is None else _f64(am_prev)
        nb, mk = 0, None
        if in_masks is not None and len(in_masks):
            mk = np.ascontiguousarray(in_masks, dtype=np.uint8)
            nb = mk.shape[0]
        cov = np.empty((d, d)); am = np.empty((d, d))
        axes = np.empty((d, d)); axes_inv = np.empty((d, d))
        lv = C.c_double(0.); rmax = C.c_double(0.); ncl = C.c_int32(0)
        self._check(self.lib.dh_friends_update(
            self.handle, _ptr(pts), n, d, 0 if kind == 'balls' else 1,
            _ptr(prev), nb, _ptr(mk), _ptr(cov), _ptr(am), _ptr(axes),
            _ptr(axes_inv), C.byref(lv), C.byref(rmax), C.byref(ncl)))
        return dict(cov=cov, am=am, axes=axes, axes_inv=axes_inv,
                    logvol=lv.value, rmax=rmax.value, nclusters=ncl.value)

    def friends_within(self, ctrs, kind, axes_inv, x, want_bits=False):
        """Counts (and optionally index bit rows) of the balls / cubes that
        contain each row of x (dh_friends_within)."""
        c = _f64(ctrs)
        n, d = c.shape
        xs = _f64(x).reshape(-1, d)
        m = xs.shape[0]
        counts = np.empty(m, dtype=np.int32)
        bits = np.empty((m, (n + 63) // 64), dtype=np.uint64) if want_bits \
            else None
        self._check(self.lib.dh_friends_within(
            self.handle, _ptr(c), n, d, 0 if kind == 'balls' else 1,
            _ptr(_f64(axes_inv)), _ptr(xs), m, _ptr(counts), _ptr(bits)))
        return counts, bits

    def friends_draw(self, state6, nsamp, ctrs, kind, axes, axes_inv,
                     return_q=False):
        """Bound.samples from one generator state (dh_friends_draw)."""
        c = _f64(ctrs)
        n, d = c.shape
        st = np.ascontiguousarray(state6, dtype=np.uint64)
        xs = np.empty((nsamp, d))
        qs = np.empty(nsamp, dtype=np.int32)
        out = np.empty(6, dtype=np.uint64)
        self._check(self.lib.dh_friends_draw(
            self.handle, _ptr(st), int(nsamp), _ptr(c), n, d,
            0 if kind == 'balls' else 1, _ptr(_f64(axes)),
            _ptr(_f64(axes_inv)), 1 if return_q else 0, _ptr(xs), _ptr(qs),
            _ptr(out)))
        return xs, qs, out

    def bound_draw(self, state4, nsamp, ctrs, axes, ams=None, logvol_ells=None,
                   return_q=False):
        """Bound.samples from one generator state (dh_bound_draw)."""
        c = _f64(ctrs)
        if c.ndim == 1:
            c = c[None, :]
        m, d = c.shape
        ax = _f64(axes).reshape(m, d, d)
        am = cp = None
        if m > 1:
            am = _f64(ams).reshape(m, d, d)
            cp = cumprob_of(logvol_ells)
        st = np.ascontiguousarray(state4, dtype=np.uint64)
        xs = np.empty((nsamp, d))
        idxs = np.empty(nsamp, dtype=np.int32)
        qs = np.empty(nsamp, dtype=np.int32)
        out = np.empty(4, dtype=np.uint64)
        self._check(self.lib.dh_bound_draw(
            self.handle, _ptr(st), int(nsamp), d, m, _ptr(c), _ptr(ax),
            _ptr(am), _ptr(cp), 1 if return_q else 0, _ptr(xs), _ptr(idxs),
            _ptr(qs), _ptr(out)))
        return xs, idxs, qs, out


def cumprob_of(logvol_ells):
    """cumsum(exp(logvol_ells - logsumexp(logvol_ells))) exactly as
    rand_choice sees it (bounding.py:543, 1300-1308)."""
    from scipy.special import logsumexp
    lv = np.asarray(logvol_ells, dtype=np.float64)
    return np.ascontiguousarray(np.cumsum(np.exp(lv - logsumexp(lv))))


_default_ctx = {}


def default_context(device=0):
    """Process-wide context per device (created lazily)."""
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.handle is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
