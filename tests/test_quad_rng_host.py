"""Host restatement of the wave-per-stream draw logic of dynesty_amd/csrc/walkq.hip (`wavegen_round` /
`wavegen_fill`, round 4): the 64 lanes of a wavefront evaluate 64 consecutive positions of ONE walker's PCG64
stream (state at position l + 1 = S_0 + G_{l+1} d with d = S_1 - S_0), scalar code resolves which role each
position plays in the sequential algorithm (normal candidate / wedge uniform of a missed candidate / the step's
uniform) from the ballot of the fast-accept test and of every lane's wedge verdict, a missed candidate takes the
next position as its wedge uniform and the later roles move up, item offsets are a masked bit count,
position 64 only yields the next round's d, the tail is finished sequentially, items go to a ring of a whole
number of steps.  The
claim pinned here: the items produced per step (nc normals, one uniform -- what propose_ball_point / randsphere
draw, internal_samplers.py:1007-1021, bounding.py:1291-1295) and the generator state left behind are EXACTLY
numpy's, step after step.  The device code is held to the oracle's walkers by the `-m gpu` tests; this test
pins the algorithm (and its constants) where no GPU exists."""
import math

import numpy as np
import pytest

from test_rng_host import M128, M64, MULT, Pcg, load_tables, normal, R, INVR

G = [0]
for _ in range(64):
    G.append((G[-1] * MULT + 1) & M128)


def ring_cap(n1):
    """a whole number of steps that holds the at most n1 - 1 unread items + the 63 a round can add"""
    return n1 * -(-(n1 + 62) // n1)


def out64(s):
    hi, lo = s >> 64, s & M64
    x, r = hi ^ lo, s >> 122
    return ((x >> r) | (x << ((64 - r) & 63))) & M64


def u53(r):
    return (r >> 11) * (1.0 / 9007199254740992.0)


class WaveGen:
    """One walker's generator as the wavefront runs it.  ring / W / c / wm are the kernel's."""

    def __init__(self, base, inc, nc, steps):
        self.inc, self.n, self.n1 = inc, nc, nc + 1
        self.S0 = base
        self.D = ((MULT - 1) * base + inc) & M128
        self.T = steps * self.n1
        self.U0 = sum(1 << b for b in range(0, 64, self.n1))
        self.cap = ring_cap(self.n1)
        self.ring = [None] * self.cap
        self.W = 0
        self.rounds = 0

    def round(self, ki, wi, fi):
        self.rounds += 1
        n, n1 = self.n, self.n1
        W = self.W
        c = W - ((W * ((1 << 32) // n1 + 1)) >> 32) * n1
        CAP = self.cap
        wm = W - ((W * ((1 << 32) // CAP + 1)) >> 32) * CAP
        assert c == W % n1 and wm == W % CAP
        st = [(self.S0 + G[l + 1] * self.D) & M128 for l in range(64)]
        r = [out64(s) for s in st]
        idx = [v & 0xff for v in r]
        rabs = [(v >> 9) & 0x000fffffffffffff for v in r]
        x = [(-1.0 if v & 0x100 else 1.0) * (ra * wi[i]) for v, ra, i in zip(r, rabs, idx)]
        missmask = sum((0 if rabs[l] < ki[idx[l]] else 1) << l for l in range(63))
        umask = (self.U0 << (n - c)) & M64
        dead, endpos, tailf = 0, 63, -1
        m = missmask & ~umask
        if m:
            # every lane's wedge verdict with the next lane's uniform (lane 63 never asks)
            take = [False] * 64
            for l in range(63):
                ic = max(idx[l], 1)
                take[l] = (fi[ic - 1] - fi[ic]) * u53(r[l + 1]) + fi[ic] < math.exp(-0.5 * x[l] * x[l])
            while m:
                f = (m & -m).bit_length() - 1
                if idx[f] == 0:
                    tailf = endpos = f
                    break
                if f == 62:
                    endpos = 62
                    break
                acc = 1 if take[f] else 0
                dead |= (2 | (acc ^ 1)) << f
                keep = f + 2
                low = umask & ((1 << keep) - 1)
                umask = (low | ((umask >> (f + acc)) << keep)) & M64
                m = missmask & ~umask & ((M64 << keep) & M64)
        off = [l - bin(dead & ((1 << l) - 1)).count("1") for l in range(64)]
        total = endpos - bin(dead).count("1")
        need = self.T - W
        if total >= need:
            endpos = [l for l in range(64) if not (dead >> l) & 1 and off[l] == need - 1][0] + 1
            total, tailf = need, -1
        for l in range(64):
            if not (dead >> l) & 1 and off[l] < total:
                q = (wm + off[l]) % CAP
                assert self.ring[q] is None, "ring overflow"
                self.ring[q] = u53(r[l]) if (umask >> l) & 1 else x[l]
        if tailf >= 0:
            g = Pcg(st[tailf], self.inc)
            while True:
                xx = -INVR * math.log1p(-g.dbl())
                yy = -math.log1p(-g.dbl())
                if yy + yy > xx * xx:
                    q = (wm + total) % CAP
                    assert self.ring[q] is None
                    self.ring[q] = -(R + xx) if ((rabs[tailf] >> 8) & 1) else R + xx
                    total += 1
                    break
            self.S0 = g.state
            self.D = ((MULT - 1) * g.state + self.inc) & M128
        else:
            assert 1 <= endpos <= 63
            self.S0 = st[endpos - 1]
            self.D = (st[endpos] - st[endpos - 1]) & M128
        self.W = W + total

    def step_items(self, step, ki, wi, fi):
        """what the walker's four sub-lanes read at `step` (after wavegen_fill up to (step + 1) * n1)"""
        while self.W < (step + 1) * self.n1:
            self.round(ki, wi, fi)
        start = (step * self.n1) % self.cap
        assert start + self.n1 <= self.cap  # a step's items never wrap
        items = []
        for e in range(self.n1):
            q = start + e
            items.append(self.ring[q])
            self.ring[q] = None  # read: the slot is free again
        return items


def np_state(bg):
    st = bg.state["state"]
    return st["state"], st["inc"]


@pytest.mark.parametrize("nc", [2, 3, 4, 5, 7, 8, 9, 12, 16, 25, 28, 31, 32])
def test_wave_draws_equal_numpy(nc):
    ki, wi, fi = load_tables()
    for seed in range(40):
        bg = np.random.PCG64(1000 * nc + seed)
        gen = np.random.Generator(bg)
        base, inc = np_state(bg)
        steps = 30
        q = WaveGen(base, inc, nc, steps)
        for step in range(steps):
            items = q.step_items(step, ki, wi, fi)
            ref_n = gen.standard_normal(nc)
            ref_u = gen.random()
            assert items[:nc] == ref_n.tolist(), (nc, seed, step)
            assert items[nc] == ref_u
        # every item drawn: S_0 is the generator's state after the last of them
        assert q.W == q.T and q.S0 == np_state(bg)[0]


def test_wave_tail_and_wedge_paths_are_reached():
    """Long streams: wedge rejections, wedge accepts and the idx == 0 tail all occur and stay in step
    with the sequential algorithm (restated in test_rng_host.normal, itself pinned to numpy)."""
    ki, wi, fi = load_tables()
    bg = np.random.PCG64(7)
    base, inc = np_state(bg)
    steps = 20000
    q, g = WaveGen(base, inc, 25, steps), Pcg(base, inc)
    ntail = 0
    for step in range(steps):
        items = q.step_items(step, ki, wi, fi)
        for i in range(25):
            s0 = g.state
            x = normal(g, ki, wi, fi)
            assert items[i] == x
            r0 = out64((s0 * MULT + inc) & M128)
            if (r0 & 0xff) == 0 and not ((r0 >> 9) & 0x000fffffffffffff) < ki[0]:
                ntail += 1
        assert items[25] == g.dbl()
    assert q.S0 == g.state
    assert ntail >= 1
    # 26 items a step, at most 63 positions a round, ~1.2 % of the candidates take a second position
    assert 0.41 < q.rounds / steps < 0.45


def test_jump_constants_match_the_kernel_source():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynesty_amd", "csrc",
                            "walkq.hip")).read()
    hi = int(re.search(r"DH_PCG_MULTM1_HI 0x([0-9a-f]+)ull", src).group(1), 16)
    lo = int(re.search(r"DH_PCG_MULTM1_LO 0x([0-9a-f]+)ull", src).group(1), 16)
    assert ((hi << 64) | lo) == MULT - 1
    assert "return n1 * ((n1 + 62 + n1 - 1) / n1);" in src  # ring_cap
