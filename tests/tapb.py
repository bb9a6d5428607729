"""Tap B without dynesty on the GPU box: record, in the build container, every call the REAL dynesty
NestedSampler makes into the backend through the drop-in plugin objects (arguments only), and replay
the recording elsewhere against any backend.

  RecordingBackend(inner)   proxy around a backend; logs (method, args, kwargs) of every call
  save_trace / load_trace   one .npz: arrays + a JSON manifest
  replay(trace, backend)    calls the backend with the recorded arguments, returns the list of results
  compare(name, a, b)       method-aware comparison of two results (indices / counters exact, coordinates
                            and ellipsoids to the tolerances of the device parity tests)

The oracle backend (tests/oracle_backend.py, NumPy) travels with the repo, so the replay can compute the
expected results on the spot; the trace holds inputs only.  Test infrastructure (nothing here is imported
by the product).
"""
import json
import time

import numpy as np

import inputs


class RecordingBackend:
    def __init__(self, inner):
        self._inner = inner
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self._inner, name)
        if not callable(fn):
            return fn

        def wrapped(*args, **kwargs):
            self.calls.append((name, _freeze(args), _freeze(kwargs)))
            return fn(*args, **kwargs)
        return wrapped


def _freeze(x):
    """Deep copy of an argument tree into plain containers (arrays copied: some calls mutate in place)."""
    from dynesty_amd import problems
    if isinstance(x, problems.Problem):
        return {"__problem__": x.name}
    if isinstance(x, np.ndarray):
        return x.copy()
    if isinstance(x, (list, tuple)):
        return [_freeze(v) for v in x]
    if isinstance(x, dict):
        return {k: _freeze(v) for k, v in x.items()}
    if isinstance(x, (np.generic,)):
        return x.item()
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    raise TypeError(f"cannot record argument of type {type(x)}")


def save_trace(path, calls, meta=None):
    arrays, manifest, seen = {}, [], {}

    def enc(x):
        if isinstance(x, np.ndarray):
            # identical arrays are stored once (a bound's ctrs / ams travel with every single-point
            # contains() of a queue fill)
            h = (x.shape, x.dtype.str, x.tobytes())
            key = seen.get(h)
            if key is None:
                key = f"a{len(arrays)}"
                arrays[key] = x
                seen[h] = key
            return {"__array__": key}
        if isinstance(x, list):
            return [enc(v) for v in x]
        if isinstance(x, dict):
            return {k: enc(v) for k, v in x.items()}
        return x
    for name, args, kwargs in calls:
        manifest.append([name, enc(args), enc(kwargs)])
    arrays["manifest"] = np.frombuffer(json.dumps({"meta": meta or {}, "calls": manifest}).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)


def load_trace(path):
    z = np.load(path)
    doc = json.loads(bytes(z["manifest"]).decode())

    def dec(x):
        if isinstance(x, dict):
            if "__array__" in x:
                return z[x["__array__"]]
            if "__problem__" in x:
                return _problem(x["__problem__"])
            return {k: dec(v) for k, v in x.items()}
        if isinstance(x, list):
            return [dec(v) for v in x]
        return x
    return doc["meta"], [(name, dec(args), dec(kwargs)) for name, args, kwargs in doc["calls"]]


_probs = {}


def _problem(name):
    if name not in _probs:
        _probs[name] = inputs.problem(name)
    return _probs[name]


# in-place contract of scale_to_logvol(covs, ams, axes, axlens, logvols, targets): the first five mutate
_INPLACE = {"scale_to_logvol": 5}


def replay(calls, backend, timer=None):
    """Results of every recorded call on `backend` (arguments are copied first: some calls mutate them).
    timer: optional dict name -> accumulated seconds."""
    out = []
    for name, args, kwargs in calls:
        a = [np.array(v) if isinstance(v, np.ndarray) else (list(v) if isinstance(v, list) else v) for v in args]
        kw = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in kwargs.items()}
        t0 = time.perf_counter()
        try:
            r = getattr(backend, name)(*a, **kw)
        except (ValueError, RuntimeError) as exc:  # reference error paths are part of the contract
            r = ("raised", type(exc).__name__)
        dt = time.perf_counter() - t0
        if timer is not None:
            timer[name] = timer.get(name, 0.0) + dt
        if name in _INPLACE and not (isinstance(r, tuple) and r and r[0] == "raised"):
            r = tuple(a[:_INPLACE[name]])
        out.append(r)
    return out


def _cmp(path, a, b, ftol):
    if isinstance(a, dict):
        skip = {"nnodes", "labels"}  # device diagnostics of rebuild(): no reference analogue
        assert isinstance(b, dict) and set(a) - skip == set(b) - skip, (path, set(a) ^ set(b))
        for k in a:
            if k not in skip:
                _cmp(f"{path}.{k}", a[k], b[k], ftol)
    elif isinstance(a, (tuple, list)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _cmp(f"{path}[{i}]", x, y, ftol)
    elif a is None or b is None:
        assert a is None and b is None, path
    else:
        x, y = np.asarray(a), np.asarray(b)
        assert x.shape == y.shape, (path, x.shape, y.shape)
        if x.dtype.kind in "iub" or y.dtype.kind in "iub" or x.dtype.kind in "US":
            np.testing.assert_array_equal(x, y, err_msg=path)
        else:
            rtol, atol_rel = ftol(path)
            scale = float(np.max(np.abs(y))) if y.size else 0.0
            np.testing.assert_allclose(x, y, rtol=rtol, atol=atol_rel * max(scale, 1e-300), err_msg=path)


def compare(name, got, want):
    """Hold `got` (device) to `want` (oracle backend) for one call of method `name`."""
    def ftol(path):
        leaf = path.split(".")[-1].split("[")[0]
        if name in ("rwalk_batch", "slice_batch", "unif_batch", "unif_friends_batch", "bound_draw",
                    "rwalk_propose", "unif_propose", "friends_draw", "slice_feed"):
            return (1e-11, 1e-12) if leaf == "logl" else (0.0, 2e-12)   # coordinates: 1e-12 of the unit cube
        if name in ("rebuild", "rebuild_many", "ell_from_cov", "scale_to_logvol"):
            if leaf in ("ams",) or path.endswith("[2]") and name in ("ell_from_cov",):
                return (0.0, 1e-8)
            return (1e-9, 1e-9)
        if name == "contains":
            return (1e-9, 1e-12)
        return (1e-9, 1e-12)
    _cmp(name, got, want, ftol)
