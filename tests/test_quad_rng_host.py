"""Host restatement of the four-lanes-per-walker draw logic of dynesty_amd/csrc/walkq.hip
(`quad_draw_step`): one PCG64 stream consumed by four lanes that each hold the state t + 1 steps ahead,
four ziggurat candidates classified per round, a missed candidate waiting for its wedge uniform (= sub-lane
0's candidate of the next round after one re-aligning jump), the tail finished sequentially.  The claim
pinned here: the items produced per step (nc normals, one uniform -- what propose_ball_point / randsphere
draw, internal_samplers.py:1007-1021, bounding.py:1291-1295) and the generator state left behind are
EXACTLY numpy's, step after step.  The device code is held to the oracle's walkers by the `-m gpu` tests;
this test pins the algorithm (and its jump constants) where no GPU exists."""
import math

import numpy as np
import pytest

from test_rng_host import M128, M64, MULT, Pcg, load_tables, normal, R, INVR

A = [1]
G = [0]
for _ in range(4):
    A.append((A[-1] * MULT) & M128)
    G.append((G[-1] * MULT + 1) & M128)
MULT_INV = pow(MULT, -1, 1 << 128)


def out64(s):
    hi, lo = s >> 64, s & M64
    x, r = hi ^ lo, s >> 122
    return ((x >> r) | (x << ((64 - r) & 63))) & M64


def u53(r):
    return (r >> 11) * (1.0 / 9007199254740992.0)


class Quad:
    """The four lanes of one walker.  stats counts rounds and jumps."""

    def __init__(self, base, inc):
        self.inc = inc
        self.S = [(A[t + 1] * base + G[t + 1] * inc) & M128 for t in range(4)]
        self.rounds = 0

    def base(self):
        return ((self.S[0] - self.inc) * MULT_INV) & M128

    def rejump(self, B):
        self.S = [(A[t + 1] * B + G[t + 1] * self.inc) & M128 for t in range(4)]

    def draw_step(self, nc, ki, wi, fi):
        NI = nc + 1
        items = [None] * NI
        count, pend, pidx, px = 0, False, 0, 0.0
        while count < NI:
            self.rounds += 1
            r = [out64(s) for s in self.S]
            shift = 0
            if pend:
                if (fi[pidx - 1] - fi[pidx]) * u53(r[0]) + fi[pidx] < math.exp(-0.5 * px * px):
                    items[count] = px
                    count += 1
                shift, pend = 1, False
            tend = min(4, shift + NI - count)
            fm, dec = 4, {}
            for t in range(shift, tend):
                my = count + t - shift
                idx = r[t] & 0xff
                rabs = (r[t] >> 9) & 0x000fffffffffffff
                x = rabs * wi[idx]
                if r[t] & 0x100:
                    x = -x
                dec[t] = (my, idx, rabs, x)
                if my < nc and not rabs < ki[idx] and fm == 4:
                    fm = t
            stop = min(fm, tend)
            for t in range(shift, stop):
                my, idx, rabs, x = dec[t]
                items[my] = x if my < nc else u53(r[t])
            count += stop - shift
            if fm < tend:  # a miss inside the valid range
                my, idx, rabs, x = dec[fm]
                if idx == 0:
                    g = Pcg(self.S[fm], self.inc)
                    while True:
                        xx = -INVR * math.log1p(-g.dbl())
                        yy = -math.log1p(-g.dbl())
                        if yy + yy > xx * xx:
                            items[count] = -(R + xx) if ((rabs >> 8) & 1) else R + xx
                            break
                    count += 1
                    self.rejump(g.state)
                else:
                    pend, pidx, px = True, idx, x
                    self.rejump(self.S[fm])
            elif tend < 4:
                self.rejump(self.S[tend - 1])
            else:
                self.S = [(A[4] * s + G[4] * self.inc) & M128 for s in self.S]
        assert not pend
        return items


def np_state(bg):
    st = bg.state["state"]
    return st["state"], st["inc"]


@pytest.mark.parametrize("nc", [1, 3, 4, 5, 9, 16, 25, 28, 32])
def test_quad_draws_equal_numpy(nc):
    ki, wi, fi = load_tables()
    for seed in range(40):
        bg = np.random.PCG64(1000 * nc + seed)
        gen = np.random.Generator(bg)
        base, inc = np_state(bg)
        q = Quad(base, inc)
        for step in range(30):
            items = q.draw_step(nc, ki, wi, fi)
            ref_n = gen.standard_normal(nc)
            ref_u = gen.random()
            assert items[:nc] == ref_n.tolist(), (nc, seed, step)
            assert items[nc] == ref_u
            assert q.base() == np_state(bg)[0]


def test_quad_tail_and_wedge_paths_are_reached():
    """Long streams: wedge rejections, wedge accepts and the idx == 0 tail all occur and stay in step
    with the sequential algorithm (restated in test_rng_host.normal, itself pinned to numpy)."""
    ki, wi, fi = load_tables()
    bg = np.random.PCG64(7)
    base, inc = np_state(bg)
    q, g = Quad(base, inc), Pcg(base, inc)
    ntail = 0
    for step in range(20000):
        items = q.draw_step(25, ki, wi, fi)
        for i in range(25):
            s0 = g.state
            x = normal(g, ki, wi, fi)
            assert items[i] == x
            r0 = out64((s0 * MULT + inc) & M128)
            if (r0 & 0xff) == 0 and not ((r0 >> 9) & 0x000fffffffffffff) < ki[0]:
                ntail += 1
        assert items[25] == g.dbl()
        assert q.base() == g.state
    assert ntail >= 1
    # 7 rounds without a miss; a miss costs at most one more
    assert 7.0 <= q.rounds / 20000 < 7.6


def test_jump_constants_match_the_kernel_source():
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynesty_amd", "csrc",
                            "walkq.hip")).read()

    def consts(fn):
        body = src.split(f"U128 {fn}(int j)")[1].split("return")[0]
        v = [int(x, 16) for x in re.findall(r"0x([0-9a-f]+)ull", body)]
        return [(v[2 * i] << 64) | v[2 * i + 1] for i in range(4)]
    assert consts("jump_A") == A[1:]
    assert consts("jump_G") == G[1:]
    hi = int(re.search(r"DH_PCG_MULT_INV_HI 0x([0-9a-f]+)ull", src).group(1), 16)
    lo = int(re.search(r"DH_PCG_MULT_INV_LO 0x([0-9a-f]+)ull", src).group(1), 16)
    assert ((hi << 64) | lo) == MULT_INV and (MULT_INV * MULT) & M128 == 1
