"""Host-side restatement of the RNG algorithms the device implements
(dynesty_amd/csrc/rng_pcg64.h): PCG64 stepping/output, SeedSequence child
hashing and NumPy's ziggurat with the tables shipped in
csrc/npy_ziggurat_tables.h -- checked bit-for-bit against numpy itself and the
golden stream.  This pins the *tables and the algorithm*; tests/test_gpu_rng.py
pins the device code."""
import math
import os
import re
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M128, M64 = (1 << 128) - 1, (1 << 64) - 1
MULT = 0x2360ED051FC65DA44385DF649FCCF645
R, INVR = 3.6541528853610087963519472518, 0.27366123732975827203338247596


def load_tables():
    txt = open(os.path.join(ROOT, "dynesty_amd", "csrc",
                            "npy_ziggurat_tables.h")).read()

    def arr(name):
        body = txt.split(name)[1].split('{')[1].split('}')[0]
        return [int(x, 16) for x in re.findall(r'0x([0-9A-F]+)ULL', body)]

    def dbl(x):
        return struct.unpack('<d', struct.pack('<Q', x))[0]

    return (arr('dh_zig_ki_host'), [dbl(x) for x in arr('dh_zig_wi_bits_host')],
            [dbl(x) for x in arr('dh_zig_fi_bits_host')])


class Pcg:

    def __init__(self, state, inc):
        self.state, self.inc = state, inc

    def next64(self):
        self.state = (self.state * MULT + self.inc) & M128
        hi, lo = self.state >> 64, self.state & M64
        x, r = hi ^ lo, self.state >> 122
        return ((x >> r) | (x << ((64 - r) & 63))) & M64

    def dbl(self):
        return (self.next64() >> 11) * (1.0 / 9007199254740992.0)


def normal(g, ki, wi, fi):
    while True:
        r = g.next64()
        idx = r & 0xff
        r >>= 8
        sign = r & 1
        rabs = (r >> 1) & 0x000fffffffffffff
        x = rabs * wi[idx]
        if sign:
            x = -x
        if rabs < ki[idx]:
            return x
        if idx == 0:
            while True:
                xx = -INVR * math.log1p(-g.dbl())
                yy = -math.log1p(-g.dbl())
                if yy + yy > xx * xx:
                    return -(R + xx) if ((rabs >> 8) & 1) else R + xx
        elif (fi[idx - 1] - fi[idx]) * g.dbl() + fi[idx] < math.exp(-0.5 * x * x):
            return x


def child_pcg(entropy_words, child):
    IA, MA, IB, MB = 0x43b0d7e5, 0x931e8875, 0x8b51f9dd, 0x58f38ded
    ML, MR, m32 = 0xca01f9dd, 0x4973f715, 0xffffffff
    ent = [int(x) for x in entropy_words]
    ent += [0] * max(0, 4 - len(ent))
    ent.append(child)
    hc = [IA]

    def hashmix(v):
        v ^= hc[0]
        hc[0] = (hc[0] * MA) & m32
        v = (v * hc[0]) & m32
        return v ^ (v >> 16)

    def mix(x, y):
        r = (ML * x - MR * y) & m32
        return r ^ (r >> 16)

    pool = [hashmix(ent[i]) for i in range(4)]
    for s in range(4):
        for d in range(4):
            if s != d:
                pool[d] = mix(pool[d], hashmix(pool[s]))
    for s in range(4, len(ent)):
        for d in range(4):
            pool[d] = mix(pool[d], hashmix(ent[s]))
    h, w = IB, []
    for i in range(8):
        v = pool[i % 4] ^ h
        h = (h * MB) & m32
        v = (v * h) & m32
        w.append(v ^ (v >> 16))
    q = [w[2 * i] | (w[2 * i + 1] << 32) for i in range(4)]
    initstate, initseq = (q[0] << 64) | q[1], (q[2] << 64) | q[3]
    inc = ((initseq << 1) | 1) & M128
    st = inc  # (0 * MULT + inc)
    st = (st + initstate) & M128
    st = (st * MULT + inc) & M128
    return Pcg(st, inc)


def test_seedsequence_and_stream_golden(golden_rng):
    from dynesty_amd import _lib
    g = golden_rng
    words = _lib.entropy_words(g["ss/entropy"])
    for i in range(5):
        p = child_pcg(words, i)
        want = tuple(int(x) for x in g["ss/pcg_state"][i])
        assert (p.state >> 64, p.state & M64, p.inc >> 64, p.inc & M64) == want
    ki, wi, fi = load_tables()
    p = child_pcg(words, 2)
    got = np.array([normal(p, ki, wi, fi) for _ in range(5000)])
    np.testing.assert_array_equal(got, g["stream/normals"])
    np.testing.assert_array_equal(np.array([p.dbl() for _ in range(100)]),
                                  g["stream/uniforms"])


def test_dynesty_style_entropy_vs_numpy():
    from dynesty_amd import _lib
    rs = np.random.default_rng(5)
    ent = rs.integers(0, 2**63 - 1, size=4)  # utils.py:1007
    kids = np.random.SeedSequence(ent).spawn(4)
    for i, c in enumerate(kids):
        s = np.random.PCG64(c).state["state"]
        p = child_pcg(_lib.entropy_words(ent), i)
        assert p.state == s["state"] and p.inc == s["inc"]
    # state-word helpers round-trip and keep numpy's buffered uint32 half
    bg = np.random.PCG64(9)
    gen = np.random.Generator(bg)
    gen.integers(10)  # leaves has_uint32 = 1
    before = bg.state
    _lib.set_pcg_state_words(bg, _lib.pcg_state_words(bg))
    assert bg.state == before and before["has_uint32"] == 1


def test_ziggurat_tail_vs_numpy():
    ki, wi, fi = load_tables()
    bg = np.random.PCG64()
    st = bg.state
    st["state"] = {"state": 12345, "inc": (6789 << 1) | 1}
    bg.state = st
    want = np.random.Generator(bg).standard_normal(120000)
    p = Pcg(12345, (6789 << 1) | 1)
    got = np.array([normal(p, ki, wi, fi) for _ in range(120000)])
    assert np.abs(want).max() > R  # the tail branch was exercised
    np.testing.assert_array_equal(got, want)


def test_frames_dedupe_by_memory_and_value():
    """samplers._frames: HipMultiEllipsoid.get_random_axes returns a fresh view of
    axes_ells[i] per call -- a fill over nells ellipsoids must upload <= nells frames."""
    from types import SimpleNamespace
    from dynesty_amd import samplers
    from dynesty_amd.bounding import HipMultiEllipsoid
    rng = np.random.default_rng(3)
    nells, d = 3, 4
    covs = np.array([np.eye(d) * (0.01 * (i + 1)) for i in range(nells)])
    ctrs = np.full((nells, d), 0.5) + 0.1 * np.arange(nells)[:, None]
    m = HipMultiEllipsoid.__new__(HipMultiEllipsoid)
    m.axes_ells = np.array([np.linalg.cholesky(c) for c in covs])
    m.logvol_ells = np.log(np.arange(1., nells + 1))
    m.logvol = float(np.log(np.exp(m.logvol_ells).sum()))
    args = [SimpleNamespace(axes=m.get_random_axes(rng)) for _ in range(200)]
    # (round 5) one cached view object per ellipsoid: a queue's frames deduplicate by identity alone
    assert len({id(a.axes) for a in args}) == nells
    frames, idx = samplers._frames(args)
    assert len(frames) <= nells and idx is not None
    for a, j in zip(args, idx):
        np.testing.assert_array_equal(frames[j], a.axes)
    # ... and the draws are the reference's expression on the same generator (bounding.py:726-731)
    ra, rb = np.random.default_rng(8), np.random.default_rng(8)
    probs = np.exp(m.logvol_ells - m.logvol)
    for _ in range(300):
        want = min(np.searchsorted(np.cumsum(probs), ra.random()), nells - 1)
        assert m.get_random_axes(rb) is m._axes_cache[2][want]
    # a change of the volumes (scale_to_logvol works in place) refreshes the cumulative weights
    m.logvol_ells[0] += 3.0
    m.logvol = float(np.log(np.exp(m.logvol_ells).sum()))
    probs = np.exp(m.logvol_ells - m.logvol)
    for _ in range(100):
        want = min(np.searchsorted(np.cumsum(probs), ra.random()), nells - 1)
        np.testing.assert_array_equal(m.get_random_axes(rb), m.axes_ells[want])
    # fresh views of the same memory (what the reference's own classes hand out): deduplicated by the memory they view
    args = [SimpleNamespace(axes=m.axes_ells[i % nells]) for i in range(200)]
    assert len({id(a.axes) for a in args}) == 200
    frames, idx = samplers._frames(args)
    assert len(frames) == nells
    for a, j in zip(args, idx):
        np.testing.assert_array_equal(frames[j], a.axes)
    # equal by value only (separate copies) also collapses
    args = [SimpleNamespace(axes=m.axes_ells[i % nells].copy()) for i in range(50)]
    frames, idx = samplers._frames(args)
    assert len(frames) == nells
    # a single frame -> no index array
    frames, idx = samplers._frames([SimpleNamespace(axes=m.axes_ells[0])] * 5)
    assert len(frames) == 1 and idx is None


def test_scalar_contains_matches_reference_golden():
    """HipMultiEllipsoid.contains / HipEllipsoid.contains for a single point are host scalar queries (the
    reference's einsum): held to the reference's golden verdicts without a GPU."""
    import os
    import inputs
    from dynesty_amd.bounding import HipEllipsoid, HipMultiEllipsoid
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bounding.npz"))
    for name in inputs.CLOUDS_SMALL:
        probes = g[f"{name}/kat/probes"]
        m = HipMultiEllipsoid.__new__(HipMultiEllipsoid)
        m._set_arrays(g[f"{name}/mu/ctrs"], g[f"{name}/mu/covs"], g[f"{name}/mu/ams"], g[f"{name}/mu/axes"],
                      g[f"{name}/mu/axlens"], g[f"{name}/mu/logvol_ells"])
        np.testing.assert_array_equal([m.contains(p) for p in probes], g[f"{name}/kat/contains"])
        d = probes.shape[1]
        e = HipEllipsoid(d, ctr=g[f"{name}/be/ctr"], cov=g[f"{name}/be/cov"], am=g[f"{name}/be/am"],
                         axes=g[f"{name}/be/axes"], axlens=g[f"{name}/be/axlens"], logvol=float(g[f"{name}/be/logvol"]))
        np.testing.assert_array_equal([e.contains(p) for p in probes], g[f"{name}/single/contains"])


def test_mirror_generator_is_numpys_pcg64():
    """tests/resident_mirror.py (the host mirror of the resident loop) carries its own PCG64: next64 / next_double /
    the buffered 32-bit halves against numpy on SeedSequence children, and pcg_setseq_128_srandom_r against numpy's
    own seeding of a PCG64 from (state, inc)."""
    from resident_mirror import Pcg, child_words
    ent = [5, 8, 7]
    for child in (0, 3, 0x80000001):
        bg = np.random.PCG64(np.random.SeedSequence(ent, spawn_key=(child,)))
        g = Pcg(child_words(ent, child))
        ref = np.random.Generator(bg)
        raw = bg.random_raw(5)
        assert [g.next64() for _ in range(5)] == [int(x) for x in raw]
        np.testing.assert_array_equal([g.next_double() for _ in range(7)], ref.random(7))
    # srandom: state = 0, inc = (seq << 1) | 1, step, state += initstate, step
    g = Pcg()
    g.seed((123 << 64) | 456, (789 << 64) | 1011)
    mult = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645
    m128 = (1 << 128) - 1
    inc = ((((789 << 64) | 1011) << 1) | 1) & m128
    st = (0 * mult + inc) & m128
    st = (st + ((123 << 64) | 456)) & m128
    st = (st * mult + inc) & m128
    assert g.state == st and g.inc == inc
    # random_interval(max) on the buffered 32-bit stream: masked rejection, both halves of a 64-bit draw used
    g = Pcg(child_words(ent, 9))
    h = Pcg(child_words(ent, 9))
    draws = [g.interval(99) for _ in range(200)]
    assert all(0 <= d <= 99 for d in draws) and len(set(draws)) > 60
    v = h.next64()
    first = [x & 127 for x in (v & 0xFFFFFFFF, v >> 32)]
    got = [d for d in first if d <= 99]
    assert draws[:len(got)] == got
