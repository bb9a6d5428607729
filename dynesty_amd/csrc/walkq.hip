// rwalk with FOUR lanes per walker (16 walkers per wavefront), frame product and Gaussian
// quadratic form on the fp64 matrix cores.
//
// Why: with one walker per lane (walk.hip) the launch that the evidence
// gate allows -- 64 runs x 512 walkers in flight -- is 512 wavefronts for 1024 SIMDs, each holding a
// 45-step dependent chain: half the chip idles and the other half is latency-bound.  Here a walker is
// spread over the four lanes {j, j+16, j+32, j+48} of a wavefront (j = walker within the wave), which
// is exactly the operand layout of v_mfma_f64_16x16x4_f64:
//
//   element e of any D-vector of the walker (u, u', dr, v, P v) lives in sub-lane t = e & 3,
//   register e >> 2                                     (B operand: k = lane >> 4, column = lane & 15;
//                                                        result rows (lane >> 4) + 4 r: the same map)
//
// so du = axes . dr and w = P . v of all 16 walkers of a wave are 2 x ceil(D / 4) matrix instructions
// each, input and output in place, the frame's fragments resident in registers, the precision matrix's in
// LDS.  Everything element-wise (step, cube check, prior) runs on a quarter of the vector per lane; sums
// over a vector are two cross-lane adds.
//
// The generator stays numpy's PCG64, consumed exactly as the sequential algorithm does
// (internal_samplers.py:1007-1021: nc normals, one uniform per step; bounding.py:1291-1295).  The 128-bit
// LCG jumps (state after j steps = A_j s + G_j inc), so sub-lane t holds the state t + 1 steps ahead and a
// round classifies four ziggurat candidates at once: candidates in front of the first one that misses the
// fast accept are taken as they are; the missed one waits for its wedge uniform, which is the NEXT draw
// and therefore sub-lane 0's candidate of the next round (re-aligned by one jump -- the multiply every
// round does anyway); the rare tail case is finished sequentially.  tests/test_quad_rng_host.py restates
// the round logic on the host and holds it to numpy draw for draw.
#include <stdlib.h>

#include <hiprand/hiprand_kernel.h>

#include "ctx.h"
#include "ns_sort.h"
#include "rng_pcg64.h"

using namespace dh;

namespace {

typedef double mfma_acc __attribute__((ext_vector_type(4)));
#define DH_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ITEMS: the PCG64 stream written out by itemgen_kernel; ITEMS32: the Philox stream written out by philox_items_kernel
enum : int { RNGQ_PCG64 = 0, RNGQ_PHILOX = 1, RNGQ_ITEMS = 2, RNGQ_ITEMS32 = 3 };

struct RwalkQArgs {
  ProblemDev prob;
  int k, ndim, walks, m;
  double scale, loglstar;
  const double* u0;
  const double* axes;  // m frames, row-major ndim x ndim, column i = axis i (as the caller holds them)
  const int32_t* axes_idx;
  const uint64_t* rng_in;
  double* u;
  double* v;
  double* logl;
  int32_t* nacc;
  int32_t* nrej;
  uint64_t* rng_out;
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  const uint64_t* pcg_jump;
  const double* items;  // RNGQ_ITEMS: [walker][walks * (ndim + 1)]
  const float* items32;  // RNGQ_ITEMS32: [walker][walks][4 (nb + 1)] (philox_items_kernel)
  int wbase;            // index of this launch's first walker in the caller's batch (run lookup)
  const double* run_loglstar;
  const double* run_scale;
  const int* run_mode;
  int wpr, my_mode;
  unsigned long long ph_seed, ph_seq0, ph_offset;
  const int8_t* bc;  // DH_BC_* per dimension, or null: every coordinate hard
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// value of the same walker's sub-lane `st` (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ uint32_t grp32(uint32_t x, int srclane) { return (uint32_t)__shfl((int)x, srclane); }
__device__ __forceinline__ U128 grp128(const U128& s, int srclane) {
  U128 r;
  r.hi = ((uint64_t)grp32((uint32_t)(s.hi >> 32), srclane) << 32) | grp32((uint32_t)s.hi, srclane);
  r.lo = ((uint64_t)grp32((uint32_t)(s.lo >> 32), srclane) << 32) | grp32((uint32_t)s.lo, srclane);
  return r;
}
// sum / product over the four sub-lanes of a walker; every sub-lane gets the same bits
__device__ __forceinline__ double grp_sum(double x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ double grp_prod(double x) {
  x *= __shfl_xor(x, 16);
  x *= __shfl_xor(x, 32);
  return x;
}
// bits j, j+16, j+32, j+48 of a wave ballot, moved to bits 0, 16, 32, 48
__device__ __forceinline__ uint64_t grp_bits(uint64_t ballot, int j) {
  return (ballot >> j) & 0x0001000100010001ull;
}
__device__ __forceinline__ int grp_first(uint64_t g) {  // first sub-lane whose bit is set, 4 if none
  return g ? (__ffsll((long long)g) - 1) >> 4 : 4;
}

#define DH_PCG_MULTM1_HI 0x2360ed051fc65da4ull  // mult - 1
#define DH_PCG_MULTM1_LO 0x4385df649fccf644ull

__device__ __forceinline__ U128 sub128(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo - b.lo;
  r.hi = a.hi - b.hi - (a.lo < b.lo ? 1ull : 0ull);
  return r;
}

// The ziggurat's two per-draw look-ups (acceptance bound ki, scale wi) side by side: one 16-byte LDS read per draw
// instead of two 8-byte reads at random banks; fi (wedge test only) apart.
struct ZigQ {
  ulonglong2 kw[256];  // .x = ki, .y = bits of wi
  float2 ff[256];      // single-precision (fi[i-1] - fi[i], fi[i]) for the first look at a wedge test
};

// ---- one PCG64 stream per walker, drawn by the whole wavefront ----------------------------------------------
// Round 4.  The four-lanes-per-walker draw logic of round 3 (every quad classifying four candidates of its own
// stream per round, lane exchanges of 128-bit state after every miss and at every step end) was 64 % of the
// kernel.  The stream a walker consumes is a function of its generator alone (internal_samplers.py:1007-1021:
// per step n normals and one uniform, whatever the walk accepts), so the wavefront now serves its sixteen
// walkers ONE AFTER THE OTHER, all 64 lanes on one stream: with d = S_1 - S_0 the LCG's state after k steps is
// S_0 + G_k d (G_k = 1 + mult + ... + mult^(k-1)), so lane l evaluates position l + 1 with one 128-bit
// multiply by its own constant, and a round classifies 63 consecutive positions (the 64th only yields the
// next round's d = S_64 - S_63: no scalar multiply anywhere).  What role a position plays -- normal candidate,
// wedge uniform of a missed candidate, the step's uniform -- is the sequential algorithm's, resolved by scalar
// code on the ballot of the fast-accept test (a walker is wave-uniform here); a missed candidate takes the next
// position as its wedge uniform and the round goes on behind it.  Items leave in consumption order into the
// walker's ring in LDS (96 entries: at most 32 unread + 63 new), from which the four sub-lanes of the walker
// pick a step's n + 1 values.  tests/test_quad_rng_host.py restates the round on the host and holds it to
// numpy draw for draw; tests/test_gpu_rwalkq.py holds the kernel to the oracle's walkers.
// Ring geometry: the capacity is a whole number of steps (cap = n1 * ceil((n1 + 62) / n1) >= the at most n1 - 1
// unread items + the 63 a round can add), so that a step's n1 items never wrap; the row stride (doubles) is the
// largest capacity of the kernel's dimensions made 2 (mod 4), which spreads 16 walkers x 4 sub-lanes over the banks.
__host__ __device__ constexpr int ring_cap(int n1) { return n1 * ((n1 + 62 + n1 - 1) / n1); }
template <int NR>
struct RingGeom {
  static constexpr int lo = 4 * (NR - 1) + 2, hi = 4 * NR + 1;  // n1 = ndim + 1 of the dimensions this NR serves
  static constexpr int cmax() {
    int m = 0;
    for (int n1 = lo; n1 <= hi; ++n1) m = ring_cap(n1) > m ? ring_cap(n1) : m;
    return m;
  }
  static constexpr int stride = cmax() + ((2 - cmax() % 4) + 4) % 4;
};

#ifdef DH_WQ_PROF
#define WQP(...) __VA_ARGS__
struct WqProf { long long fill, rest, rounds, segs, wedges, t0; };
#else
#define WQP(...)
#endif

// Per-walker generator state lives in LDS (one row per walker of the wavefront): S_0, S_1 (the states at the
// next two positions' predecessors: d = S_1 - S_0) and the number of items produced.  A round reads the row of
// its walker with wave-uniform addresses (every lane gets the words: no cross-lane reads on the vector pipe) and
// the two lanes that hold the new S_0, S_1 write them back.
struct WaveGenRow {
  uint32_t s[8];  // S_0 (little-endian words 0..3), S_1 (4..7)
};
struct WaveGenLds {
  WaveGenRow row[16];
  int W[16];
};

__device__ __forceinline__ uint32_t rl32(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l) {
  return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l);
}
__device__ __forceinline__ uint32_t sfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// lane j < 16 (p = walker j's generator words): row j
__device__ __forceinline__ void wavegen_init(WaveGenLds* g, int j, const uint64_t* p) {
  const U128 base = {p[0], p[1]}, inc = {p[2], p[3]}, m = {DH_PCG_MULT_HI, DH_PCG_MULT_LO};
  const U128 s1 = add128(mul128(base, m), inc);
  uint32_t* r = g->row[j].s;
  r[0] = (uint32_t)base.lo;
  r[1] = (uint32_t)(base.lo >> 32);
  r[2] = (uint32_t)base.hi;
  r[3] = (uint32_t)(base.hi >> 32);
  r[4] = (uint32_t)s1.lo;
  r[5] = (uint32_t)(s1.lo >> 32);
  r[6] = (uint32_t)s1.hi;
  r[7] = (uint32_t)(s1.hi >> 32);
  g->W[j] = 0;
}

// wave-uniform constants of a launch
struct WaveGenConst {
  int n, n1, T, cap;
  uint32_t magic_n1, magic_cap;  // floor(2^32 / d) + 1: W / d = (W * magic) >> 32 for W < 2^32 / d
  uint64_t U0;                   // bits 0, n1, 2 n1, ... < 64
  const uint64_t* zfi;           // the ziggurat's fi table (global memory: only the rare double-precision wedge test reads it)
};

// low 128 bits of a * b in ten 32 x 32 products (the multiplier runs at a quarter of the vector rate and this
// product is the largest single item of a round; the compiler's expansion of 64-bit products takes thirteen to
// fourteen, some of them by a zero high half it does not see through)
__device__ __forceinline__ uint64_t mad_u64_u32(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
struct Limbs128 {
  uint32_t w[4];  // little-endian
};
// (opaque limbs: seen as halves of 64-bit values, their 32-bit products are widened to 64-bit ones again)
__device__ __forceinline__ Limbs128 opaque_limbs(const U128& a) {
  Limbs128 l;
  l.w[0] = (uint32_t)a.lo;
  l.w[1] = (uint32_t)(a.lo >> 32);
  l.w[2] = (uint32_t)a.hi;
  l.w[3] = (uint32_t)(a.hi >> 32);
  asm("" : "+v"(l.w[0]), "+v"(l.w[1]), "+v"(l.w[2]), "+v"(l.w[3]));
  return l;
}
__device__ __forceinline__ U128 mul128_limbs(const Limbs128& a, const U128& b) {
  const uint32_t a0 = a.w[0], a1 = a.w[1], a2 = a.w[2], a3 = a.w[3];
  uint32_t b0 = (uint32_t)b.lo, b1 = (uint32_t)(b.lo >> 32), b2 = (uint32_t)b.hi, b3 = (uint32_t)(b.hi >> 32);
  asm("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  const uint64_t p00 = mad_u64_u32(a0, b0, 0ull);
  const uint64_t p01 = mad_u64_u32(a0, b1, p00 >> 32);                     // < 2^64: (2^32 - 1)^2 + 2^32 - 1
  const uint64_t p10 = mad_u64_u32(a1, b0, (uint64_t)(uint32_t)p01);
  const uint64_t p11 = mad_u64_u32(a1, b1, (p01 >> 32) + (p10 >> 32));     // bits 64..127 of a.lo * b.lo
  uint64_t hi = mad_u64_u32(a0, b2, p11);                                  // (mod 2^64 from here on)
  hi = mad_u64_u32(a2, b0, hi);
  uint32_t top = a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0;
  asm("" : "+v"(top));
  U128 r;
  r.lo = (p10 << 32) | (uint32_t)p00;
  r.hi = hi + ((uint64_t)top << 32);
  return r;
}

// One round of walker jw's stream (jw wave-uniform): up to 63 positions -> items into ring row `rw`; returns the
// walker's new item count.
// What counts here is the NUMBER of instructions of any kind: hardware counters (profiles/r04) show a wavefront
// of this kernel issuing one instruction per ~4.4 cycles for 46 % of its time and waiting for the rest, two
// wavefronts per SIMD hiding little of each other's chains.  So the common round -- no missed candidate among its
// 63 positions, 47 % of them -- goes straight through; the wedge verdicts, first in single precision, are only
// formed when a real candidate missed.
__device__ __forceinline__ int wavegen_round(WaveGenLds* g, int jw, const Limbs128& Gl, const ZigQ* z, double* rw,
                                              int lane, const WaveGenConst& k, const uint64_t* rng_in,
                                              int walker0, int kmax WQP(, WqProf* pf)) {
#pragma clang fp contract(off)
  U128 S0, D;
  {
    const uint4 a = *reinterpret_cast<const uint4*>(g->row[jw].s), b = *reinterpret_cast<const uint4*>(g->row[jw].s + 4);
    S0.lo = ((uint64_t)a.y << 32) | a.x;
    S0.hi = ((uint64_t)a.w << 32) | a.z;
    U128 S1;
    S1.lo = ((uint64_t)b.y << 32) | b.x;
    S1.hi = ((uint64_t)b.w << 32) | b.z;
    D = sub128(S1, S0);
  }
  const uint32_t W = sfirst((uint32_t)g->W[jw]);
  const U128 st = add128(S0, mul128_limbs(Gl, D));  // state at position lane + 1
  const uint64_t r = pcg_output(st);
  const int idx = (int)(r & 0xff);
  const uint64_t rabs = (r >> 9) & 0x000fffffffffffffull;
  const double rd = __longlong_as_double((long long)(rabs | 0x4330000000000000ull)) - 4503599627370496.0;
  const ulonglong2 kw = z->kw[idx];
  double x = rd * __longlong_as_double((long long)kw.y);
  x = __longlong_as_double(__double_as_longlong(x) ^ (long long)((r & 0x100ull) << 55));
  const uint64_t missmask = __ballot(!(rabs < kw.x)) & 0x7fffffffffffffffull;  // position 63 is never consumed
  const int c = (int)(W - (uint32_t)(((uint64_t)W * k.magic_n1) >> 32) * (uint32_t)k.n1);  // item index within the step
  const int wm = (int)(W - (uint32_t)(((uint64_t)W * k.magic_cap) >> 32) * (uint32_t)k.cap);
  // roles: bit p of umask = position p is a step's uniform.  Without a miss the items follow the positions;
  // a missed candidate at f takes position f + 1 as its wedge uniform, which moves every later role up by one
  // position (candidate accepted) or two (rejected).
  uint64_t umask = k.U0 << (k.n - c);
  uint64_t dead = 0;  // positions that yield no item
  int endpos = 63, tailf = -1;
  uint64_t m = missmask & ~umask;
  WQP(++pf->rounds;)
  if (m) {
    // the wedge verdict of every lane at once, as if it were a missed candidate (numpy distributions.c:
    // (fi[idx-1] - fi[idx]) * next_double() + fi[idx] < exp(-0.5 x x)); the uniform is the next position's draw.
    // First in single precision: every term is good to 2e-7 of a value in [1e-3, 1], so outside a band of 1e-5
    // around equality the verdict is the double-precision one; a real candidate inside the band sends the
    // wavefront to the double-precision test.
    const float2 ff = z->ff[idx];
    const uint32_t nhi = (uint32_t)__shfl_down((int)(uint32_t)(r >> 40), 1);
    const float xf = (float)x;
    const float lhs = ff.x * ((float)nhi * 5.9604644775390625e-08f) + ff.y;
    const float ef = __expf(-0.5f * xf * xf);
    uint64_t accmask = __ballot(lhs < ef);
    const uint64_t zeromask = __ballot(idx == 0);
    if (__ballot(fabsf(lhs - ef) <= 1e-5f) & m) {
      const int ic = idx > 0 ? idx : 1;
      const uint32_t nlo = (uint32_t)__shfl_down((int)(uint32_t)(r >> 11), 1),
                     nh2 = (uint32_t)__shfl_down((int)(uint32_t)(r >> 43), 1);
      const double u1 = (double)(((uint64_t)nh2 << 32) | nlo) * (1.0 / 9007199254740992.0);
      const double f1 = __longlong_as_double((long long)k.zfi[ic - 1]), f0 = __longlong_as_double((long long)k.zfi[ic]);
      accmask = __ballot((f1 - f0) * u1 + f0 < exp(-0.5 * x * x));
    }
    do {
      WQP(++pf->segs;)
      const int f = (int)__ffsll((long long)m) - 1;
      if ((zeromask >> f) & 1ull) {  // tail of the distribution: finished sequentially below
        tailf = f;
        endpos = f;
        break;
      }
      if (f == 62) {  // its wedge uniform is not in this round: the next round starts with this candidate
        endpos = 62;
        break;
      }
      const uint64_t acc = (accmask >> f) & 1ull;
      dead |= (2ull | (acc ^ 1ull)) << f;
      const int keep = f + 2;
      const uint64_t low = umask & ((1ull << keep) - 1ull);
      umask = low | ((umask >> (f + (int)acc)) << keep);  // later roles move up by 2 - acc positions
      m = missmask & ~umask & (~0ull << keep);
    } while (m);
  }
  const int off = lane - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
  int total = endpos - (int)__popcll(dead);  // dead positions all lie below endpos
  const int need = k.T - (int)W;
  const bool alive = !__builtin_amdgcn_inverse_ballot_w64(dead);
  if (total >= need) {
    // the walker's last round: its last item is a step's uniform (never a miss); the generator stops there
    const uint64_t b = __ballot(alive && off == need - 1);
    endpos = (int)__ffsll((long long)b);
    total = need;
    tailf = -1;
  }
  if (alive && off < total) {
    int qi = wm + off;
    qi = qi >= k.cap ? qi - k.cap : qi;
    // a step's uniform travels as its 53 random bits; the walker's lanes scale them when they read the item
    rw[qi] = __builtin_amdgcn_inverse_ballot_w64(umask) ? __longlong_as_double((long long)(r >> 11)) : x;
  }
  if (tailf >= 0) {
    // about 3 in 10^4 draws: finished sequentially, redundantly by all lanes
    int wi = walker0 + jw;
    wi = wi < kmax ? wi : kmax - 1;
    Pcg64 s;
    s.state.hi = rl64(st.hi, tailf);
    s.state.lo = rl64(st.lo, tailf);
    s.inc.hi = rng_in[(size_t)wi * 4 + 2];
    s.inc.lo = rng_in[(size_t)wi * 4 + 3];
    const uint64_t rabsf = rl64(rabs, tailf);
    double xf;
    for (;;) {
      const double xx = -DH_ZIG_INV_R * log1p(-s.next_double());
      const double yy = -log1p(-s.next_double());
      if (yy + yy > xx * xx) {
        xf = ((rabsf >> 8) & 1) ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
        break;
      }
    }
    int qi = wm + total;
    qi = qi >= k.cap ? qi - k.cap : qi;
    ++total;
    const U128 s0n = s.state;
    s.step();
    if (lane == 0) {
      rw[qi] = xf;
      uint32_t* o = g->row[jw].s;
      o[0] = (uint32_t)s0n.lo;
      o[1] = (uint32_t)(s0n.lo >> 32);
      o[2] = (uint32_t)s0n.hi;
      o[3] = (uint32_t)(s0n.hi >> 32);
      o[4] = (uint32_t)s.state.lo;
      o[5] = (uint32_t)(s.state.lo >> 32);
      o[6] = (uint32_t)s.state.hi;
      o[7] = (uint32_t)(s.state.hi >> 32);
    }
  } else {
    // positions [0, endpos) are consumed (1 <= endpos <= 63): the states of lanes endpos - 1 and endpos are the
    // walker's new S_0 and S_1
    const int which = lane - (endpos - 1);
    if ((unsigned)which < 2u)
      *reinterpret_cast<uint4*>(g->row[jw].s + 4 * which) =
          make_uint4((uint32_t)st.lo, (uint32_t)(st.lo >> 32), (uint32_t)st.hi, (uint32_t)(st.hi >> 32));
  }
  if (lane == 0) g->W[jw] = (int)W + total;
  return (int)W + total;
}

// all sixteen walkers of the wavefront up to `target` items (their rows: ring + j * stride).  Returns when no
// walker is short any more; a round that leaves its walker short (many misses, a tail) is rare, so the walkers
// are only asked again when one reported it.
template <int STRIDE>
__device__ __forceinline__ void wavegen_fill(WaveGenLds* g, int target, const Limbs128& Gl, const ZigQ* z, double* ring,
                                             int lane, const WaveGenConst& k, const uint64_t* rng_in, int walker0,
                                             int kmax WQP(, WqProf* pf)) {
  for (;;) {
    uint32_t nm = (uint32_t)__ballot(lane < 16 && g->W[lane & 15] < target);
    if (!nm) break;
    bool again = false;
    while (nm) {
      const int jw = __ffs((int)nm) - 1;
      nm &= nm - 1;
      again |= wavegen_round(g, jw, Gl, z, ring + jw * STRIDE, lane, k, rng_in, walker0, kmax WQP(, pf)) < target;
    }
    wave_sync();
    if (!again) break;
  }
}

// ---- the generator as a pass of its own (round 4) ------------------------------------------------------------
// Inside the walk kernel a round costs ~170 instructions that two wavefronts per SIMD issue at ~10 cycles each
// (profiles/r04): 70 % of the kernel.  The item stream of a walker depends on nothing but its generator, so
// `itemgen_kernel` writes it out ahead of the walk -- walks x (n normals, then the step's uniform as its 53 random
// bits) per walker, in consumption order, T = walks (n + 1) doubles per walker -- with one wavefront per walker at
// eight wavefronts per SIMD (40 registers, the walker's state in scalar registers: no state in LDS, no ring), and
// the walk kernel (RNGQ_ITEMS) reads a step's items with plain loads, one step ahead.  Same rounds, same
// arithmetic: the streams are bit for bit the fused kernel's.
struct ItemGenArgs {
  const uint64_t* rng_in;
  uint64_t* rng_out;
  double* items;  // [walker][T]
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  const uint64_t* pcg_jump;
  const int* run_mode;
  int k, n, walks, wpr, my_mode, wbase;
  // presort workgroups (dh_ctx::PresortReq): the first ps_runs workgroups of the grid
  const double* ps_keys;
  unsigned short* ps_out;
  int ps_n, ps_runs, ps_stride;
};

__device__ __forceinline__ uint64_t mad_u64_u32_s(uint32_t a, uint32_t b_uniform, uint64_t c) {
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c) : "vcc");
  return d;
}
// a (per-lane limbs) * b (wave-uniform, in scalar registers), low 128 bits
__device__ __forceinline__ U128 mul128_limbs_s(const Limbs128& a, const U128& b) {
  const uint32_t a0 = a.w[0], a1 = a.w[1], a2 = a.w[2], a3 = a.w[3];
  const uint32_t b0 = sfirst((uint32_t)b.lo), b1 = sfirst((uint32_t)(b.lo >> 32)), b2 = sfirst((uint32_t)b.hi),
                 b3 = sfirst((uint32_t)(b.hi >> 32));
  const uint64_t p00 = mad_u64_u32_s(a0, b0, 0ull);
  const uint64_t p01 = mad_u64_u32_s(a0, b1, p00 >> 32);
  const uint64_t p10 = mad_u64_u32_s(a1, b0, (uint64_t)(uint32_t)p01);
  const uint64_t p11 = mad_u64_u32_s(a1, b1, (p01 >> 32) + (p10 >> 32));
  uint64_t hi = mad_u64_u32_s(a0, b2, p11);
  hi = mad_u64_u32_s(a2, b0, hi);
  uint32_t top = a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0;
  asm("" : "+v"(top));
  U128 r;
  r.lo = (p10 << 32) | (uint32_t)p00;
  r.hi = hi + ((uint64_t)top << 32);
  return r;
}

// The two rare paths of a round are calls, not inlined code: ocml's exp (the double-precision wedge verdict, a
// candidate inside the single-precision band: ~1 round in 10^3) and log1p x 2 + a sequential 128-bit generator (the
// tail of the distribution: 3 draws in 10^4) set the kernel's register need to 96 when inlined -- five wavefronts
// per SIMD; behind calls with scalar arguments the round itself needs 40 and the kernel is what its callees and the
// call ABI's callee-saved registers need: 74 at six wavefronts per SIMD without a spill, 64 at eight with ten spilled
// registers stored and reloaded once per walker.  MEASURED (round 5): 143 us at five, at six and at eight wavefronts
// per SIMD -- occupancy is not what binds the pass -- and the eight-wavefront form's per-walker spill is 84 MB of
// scratch traffic per launch (PMC: 433 MB instead of 330).  Hence six.  What does bind it (same round, EXPERIMENTS.md
// R5.1): the in-order dependent chain of a round -- a third fewer vector instructions, half the scalar instructions of
// the common round, the new state by v_readlane instead of ds_bpermute: each within 3 us of 140; every instruction
// ADDED costs its issue time; without the resolution of missed candidates 107 us, without the stores 133 us.
__device__ __attribute__((noinline)) bool itemgen_wedge_f64(double x, double u1, double f1, double f0) {
#pragma clang fp contract(off)
  return (f1 - f0) * u1 + f0 < exp(-0.5 * x * x);
}
struct ItemTail {
  double xf;
  uint64_t s0hi, s0lo, dhi, dlo;  // the walker's S_0 behind the tail draw and d = S_1 - S_0
};
__device__ __attribute__((noinline)) ItemTail itemgen_tail(uint64_t shi, uint64_t slo, uint64_t ihi, uint64_t ilo,
                                                           uint32_t negative) {
#pragma clang fp contract(off)
  Pcg64 s;
  s.state.hi = shi;
  s.state.lo = slo;
  s.inc.hi = ihi;
  s.inc.lo = ilo;
  ItemTail o;
  for (;;) {
    const double xx = -DH_ZIG_INV_R * log1p(-s.next_double());
    const double yy = -log1p(-s.next_double());
    if (yy + yy > xx * xx) {
      o.xf = negative ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
      break;
    }
  }
  const U128 s0n = s.state;
  s.step();
  const U128 d = sub128(s.state, s0n);
  o.s0hi = s0n.hi;
  o.s0lo = s0n.lo;
  o.dhi = d.hi;
  o.dlo = d.lo;
  return o;
}

// The round's front on 32-bit limbs (round 5): a third fewer vector instructions per round than the 64-bit form --
// measured, it buys nothing (143.7 us against 143.0: the vector pipe is not what binds the pass, see above); kept
// because it is the shorter code path and frees the vector pipe for whatever shares the SIMD:
//   * S_0 and d = S_1 - S_0 are wave-uniform and live in SGPRs: the new S_0 / S_1 come from the lanes that hold them
//     by eight v_readlane, d by a scalar borrow chain (s_sub / s_subb);
//   * state = S_0 + G d: the low 128 bits of the product as limbs (six v_mad_u64_u32 + the top limb's four products),
//     added by a four-instruction carry chain instead of re-packed 64-bit halves and a carry compare;
//   * XSL-RR's 64-bit rotate by two v_alignbit_b32 and two selects (the 64-bit shifts are not full rate), the 52
//     random bits and the uniform's 53 likewise by funnel shifts.
// The arithmetic is the same integers: the same streams, bit for bit.
__device__ __forceinline__ uint32_t sel_mask(uint64_t mask, uint32_t if_set, uint32_t if_clear) {  // per lane: mask bit ? if_set : if_clear
  uint32_t r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
  return r;
}
__device__ __forceinline__ uint32_t lo32(double x) { return (uint32_t)__double_as_longlong(x); }
__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)((uint64_t)__double_as_longlong(x) >> 32); }
struct FrontOut {
  uint32_t st[4];   // state at position lane + 1, little-endian limbs
  uint32_t rlo, rhi;  // pcg_output(state)
};
// S0 and D are wave-uniform and live in SGPRs: the products take D's limbs as their (one) scalar operand, the carry
// chain S0's lowest limb (its other three ride in VGPRs: v_addc reads vcc, and vcc + an SGPR are two constant-bus reads)
__device__ __forceinline__ FrontOut itemgen_front(const Limbs128& G, const uint32_t (&S0)[4], const uint32_t (&D)[4]) {
  const uint32_t a0 = G.w[0], a1 = G.w[1], a2 = G.w[2], a3 = G.w[3];
  const uint32_t b0 = sfirst(D[0]), b1 = sfirst(D[1]), b2 = sfirst(D[2]), b3 = sfirst(D[3]);  // (folded away: D is uniform)
  const uint64_t p00 = mad_u64_u32_s(a0, b0, 0ull);
  const uint64_t t1 = mad_u64_u32_s(a0, b1, p00 >> 32);                     // < 2^64
  const uint64_t t2 = mad_u64_u32_s(a1, b0, (uint64_t)(uint32_t)t1);
  const uint64_t t3 = mad_u64_u32_s(a1, b1, (t1 >> 32) + (t2 >> 32));       // bits 64..127 of (a.lo * b.lo)
  uint64_t t5 = mad_u64_u32_s(a0, b2, t3);                                  // (mod 2^64 from here on)
  t5 = mad_u64_u32_s(a2, b0, t5);
  uint32_t top = (uint32_t)(t5 >> 32) + a0 * b3 + a1 * b2 + a2 * b1 + a3 * b0;
  const uint32_t l0 = (uint32_t)p00, l1 = (uint32_t)t2, l2 = (uint32_t)t5;
  FrontOut o;
  asm("v_add_co_u32 %0, vcc, %8, %4\n\tv_addc_co_u32 %1, vcc, %5, %9, vcc\n\tv_addc_co_u32 %2, vcc, %6, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %7, %11, vcc"
      : "=&v"(o.st[0]), "=&v"(o.st[1]), "=&v"(o.st[2]), "=&v"(o.st[3])
      : "v"(l0), "v"(l1), "v"(l2), "v"(top), "s"(sfirst(S0[0])), "v"(S0[1]), "v"(S0[2]), "v"(S0[3])
      : "vcc");
  // XSL-RR (pcg64.h): rotr64(hi ^ lo, hi >> 58)
  const uint32_t xh = o.st[3] ^ o.st[1], xl = o.st[2] ^ o.st[0], rot = o.st[3] >> 26;
  const uint32_t ra = __builtin_amdgcn_alignbit(xh, xl, rot), rb = __builtin_amdgcn_alignbit(xl, xh, rot);  // by rot mod 32
  const bool sw = rot >= 32u;
  o.rlo = sw ? rb : ra;
  o.rhi = sw ? ra : rb;
  return o;
}
// d = S_1 - S_0 of two wave-uniform states (limbs): a scalar borrow chain
__device__ __forceinline__ void sub128_limbs(const uint32_t (&A)[4], const uint32_t (&B)[4], uint32_t (&R)[4]) {
  typedef unsigned __int128 u128;
  const u128 a = ((u128)A[3] << 96) | ((u128)A[2] << 64) | ((u128)A[1] << 32) | A[0];
  const u128 b = ((u128)B[3] << 96) | ((u128)B[2] << 64) | ((u128)B[1] << 32) | B[0];
  const u128 r = a - b;
  R[0] = (uint32_t)r;
  R[1] = (uint32_t)(r >> 32);
  R[2] = (uint32_t)(r >> 64);
  R[3] = (uint32_t)(r >> 96);
}

// the slot order of one run's live points for ns_consume (see dh_ctx::PresortReq): keys into LDS, the register sort of
// ns_sort.h, the order out to global memory.  n <= 2048.
__device__ __attribute__((noinline)) void itemgen_presort(const double* keys, unsigned short* out, int n, unsigned char* smem) {
  double* skey = (double*)smem;                           // n
  unsigned short* sidx = (unsigned short*)(skey + 2048);  // P
  const int t = threadIdx.x;
  int P = 256;
  while (P < n) P <<= 1;
  for (int i0 = t; i0 < n; i0 += 8 * 256) {
    double kv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) kv[q] = keys[i0 + q * 256 < n ? i0 + q * 256 : 0];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (i0 + q * 256 < n) skey[i0 + q * 256] = kv[q];
  }
  __syncthreads();
  switch (P / 256) {
    case 1: dh_sort::sort_slots<1>(skey, sidx, n, P); break;
    case 2: dh_sort::sort_slots<2>(skey, sidx, n, P); break;
    case 4: dh_sort::sort_slots<4>(skey, sidx, n, P); break;
    default: dh_sort::sort_slots<8>(skey, sidx, n, P); break;
  }
  for (int i = t; i < P; i += 256) out[i] = sidx[i];
}

constexpr int kPresortLds = 2048 * 8 + 2048 * 2;
// PRESORT: the form with presort workgroups in front of the grid (launched only when the resident loop has asked: the
// plain form keeps its 6 KB of LDS and its 64 bytes of scratch -- with the sort's 20 KB and call frames in it the pass
// was 3 us slower for everybody)
template <bool PRESORT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) itemgen_kernel(ItemGenArgs a) {
#pragma clang fp contract(off)
  // (one buffer: a presort workgroup never touches the ziggurat tables)
  __shared__ __attribute__((aligned(16))) unsigned char ig_smem[!PRESORT || sizeof(ZigQ) > kPresortLds ? sizeof(ZigQ) : kPresortLds];
  if constexpr (PRESORT) {
    if ((int)blockIdx.x < a.ps_runs) {
      itemgen_presort(a.ps_keys + (size_t)blockIdx.x * a.ps_n, a.ps_out + (size_t)blockIdx.x * a.ps_stride, a.ps_n, ig_smem);
      return;
    }
  }
  ZigQ& zig = *reinterpret_cast<ZigQ*>(ig_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = (int)sfirst((uint32_t)(tid >> 6));
  {
    const int i = tid;
    zig.kw[i] = make_ulonglong2(a.zki[i], a.zwi[i]);
    const int im = i > 0 ? i - 1 : 0;
    const double fa = __longlong_as_double((long long)a.zfi[im]), fb = __longlong_as_double((long long)a.zfi[i]);
    zig.ff[i] = make_float2((float)(fa - fb), (float)fb);
  }
  __syncthreads();
  const ZigQ* z = &zig;
  const U128 gj = {a.pcg_jump[2 * lane], a.pcg_jump[2 * lane + 1]};  // G_{lane + 1}
  const Limbs128 Gl = opaque_limbs(gj);
  const int n = a.n, n1 = n + 1, T = a.walks * n1;
  const uint32_t magic_n1 = (uint32_t)(0x100000000ull / (uint32_t)n1) + 1u;
  uint64_t U0 = 0;
  for (int b = 0; b < 64; b += n1) U0 |= 1ull << b;
  const int ps_runs = PRESORT ? a.ps_runs : 0;
  const int nwaves = ((int)gridDim.x - ps_runs) * 4;
  for (int w = ((int)blockIdx.x - ps_runs) * 4 + wave; w < a.k; w += nwaves) {
    if (a.run_mode && a.run_mode[(a.wbase + w) / a.wpr] != a.my_mode) continue;
    const uint64_t* p = a.rng_in + (size_t)w * 4;
    const U128 S00 = {p[0], p[1]};
    const U128 inc = {p[2], p[3]};
    uint32_t S0v[4], Dv[4];  // wave-uniform (SGPRs)
    {
      const U128 mm1 = {DH_PCG_MULTM1_HI, DH_PCG_MULTM1_LO};
      const U128 D0 = add128(mul128(S00, mm1), inc);  // S_1 - S_0
      S0v[0] = sfirst((uint32_t)S00.lo);
      S0v[1] = sfirst((uint32_t)(S00.lo >> 32));
      S0v[2] = sfirst((uint32_t)S00.hi);
      S0v[3] = sfirst((uint32_t)(S00.hi >> 32));
      Dv[0] = sfirst((uint32_t)D0.lo);
      Dv[1] = sfirst((uint32_t)(D0.lo >> 32));
      Dv[2] = sfirst((uint32_t)D0.hi);
      Dv[3] = sfirst((uint32_t)(D0.hi >> 32));
    }
    double* out = a.items + (size_t)w * T;
    uint32_t W = 0;
    // A round whose single-precision wedge verdict is inside the band is run twice: the first pass forms the
    // double-precision verdicts -- from a front of their own, and then starts the round over, so that nothing but the
    // walker's constants is live across the call --, the second (have64) takes them.  The flag is read and written
    // inside that branch alone: the common path carries no test of it.
    bool have64 = false;
    uint64_t acc64 = 0;
    const int Tfast = T - 63;
    while ((int)W < T) {
      FrontOut fr;
      uint32_t rlo, rhi, rabs_lo;
      int idx;
      double x;
      uint64_t missmask, umask, m;
      // The common round (about one in two) is a loop of its own: no candidate missed and the walker needs all 63
      // positions -- lane p stores item W + p, the new S_0 / S_1 are the states of lanes 62 and 63; sixteen scalar
      // instructions and four branches where the general resolution below costs thirty-three and ten (measured: 2 us
      // of 140, EXPERIMENTS.md R5.1).  The first round that does not qualify leaves the loop with its front formed.
      for (;;) {
        fr = itemgen_front(Gl, S0v, Dv);  // state at position lane + 1 and its output
#ifdef DH_IG_EXTRA  // marginal-cost probes: sixteen extra instructions of one kind per round (tools/r5_igslope.sh)
        {
          uint32_t e0 = fr.rlo, e1 = fr.rhi, e2 = fr.st[0], e3 = fr.st[1];
          uint64_t w0 = ((uint64_t)e1 << 32) | e0, w1 = ((uint64_t)e3 << 32) | e2;
#if DH_IG_EXTRA == 1
          asm volatile("s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1\n\t"
                       "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1\n\t"
                       "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1\n\t"
                       "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\ts_add_u32 s22, s22, 1\n\ts_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");
#elif DH_IG_EXTRA == 2
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(lane));
#elif DH_IG_EXTRA == 3
#pragma unroll
          for (int e = 0; e < 8; ++e)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %2, %0\n\tv_mad_u64_u32 %1, vcc, %2, %2, %1" : "+v"(w0), "+v"(w1) : "v"(lane) : "vcc");
#elif DH_IG_EXTRA == 4
          int a62 = 62 << 2;
          asm volatile("" : "+v"(a62));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("ds_bpermute_b32 %0, %4, %0\n\tds_bpermute_b32 %1, %4, %1\n\tds_bpermute_b32 %2, %4, %2\n\tds_bpermute_b32 %3, %4, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(a62));
#elif DH_IG_EXTRA == 5
          asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"
                       "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
#elif DH_IG_EXTRA == 6
          asm volatile("s_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\t"
                       "s_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0");
#elif DH_IG_EXTRA == 7
#pragma unroll
          for (int e = 0; e < 4; ++e)
            asm volatile("v_readlane_b32 s20, %0, 5\n\tv_readlane_b32 s21, %1, 6\n\tv_readlane_b32 s22, %2, 7\n\tv_readlane_b32 s23, %3, 8"
                         : : "v"(e0), "v"(e1), "v"(e2), "v"(e3) : "s20", "s21", "s22", "s23");
#endif
          asm volatile("" : : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(w0), "v"(w1));
        }
#endif
        rlo = fr.rlo;
        rhi = fr.rhi;
        idx = (int)(rlo & 0xffu);
        // rabs = (r >> 9) & (2^52 - 1), as halves
        rabs_lo = __builtin_amdgcn_alignbit(rhi, rlo, 9);
        const uint32_t rabs_hi = (rhi >> 9) & 0xfffffu;
        const uint64_t rabs = ((uint64_t)rabs_hi << 32) | rabs_lo;
        const double rd = __longlong_as_double((long long)(((uint64_t)(rabs_hi | 0x43300000u) << 32) | rabs_lo)) - 4503599627370496.0;
        const ulonglong2 kw = z->kw[idx];
        x = rd * __longlong_as_double((long long)kw.y);
        x = __longlong_as_double(__double_as_longlong(x) ^ (long long)((uint64_t)((rlo << 23) & 0x80000000u) << 32));
        missmask = __ballot(!(rabs < kw.x)) & 0x7fffffffffffffffull;  // position 63 is never consumed
        const int c = (int)(W - (uint32_t)(((uint64_t)W * magic_n1) >> 32) * (uint32_t)n1);
        umask = U0 << (n - c);
        m = missmask & ~umask;
        if (!((int)W < Tfast)) break;
        asm volatile("" : "+s"(m));  // (two tests, two branches: merged, the compiler spends ten scalar instructions on them)
        if (m != 0) break;
        const uint32_t ulo = __builtin_amdgcn_alignbit(rhi, rlo, 11), uhi = rhi >> 11;  // the uniform's 53 bits (r >> 11)
        const double* dst = out + (W + (uint32_t)lane);
        // lanes 0..62 store: exec's top bit cleared around the store (every lane is active here)
        asm volatile("s_bitset0_b32 exec_hi, 31\n\tglobal_store_dwordx2 %0, %1, off\n\ts_bitset1_b32 exec_hi, 31"
                     :
                     : "v"(dst), "v"(((uint64_t)sel_mask(umask, uhi, hi32(x)) << 32) | sel_mask(umask, ulo, lo32(x)))
                     : "memory");
        uint32_t S1v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S0v[q] = rl32(fr.st[q], 62);
          S1v[q] = rl32(fr.st[q], 63);
        }
        sub128_limbs(S1v, S0v, Dv);
        W += 63u;
      }
      uint64_t dead = 0;
      int endpos = 63, tailf = -1;
      if (m) {  // (the same resolution as wavegen_round's)
        const float2 ff = z->ff[idx];
        const uint32_t nhi = (uint32_t)__shfl_down((int)(rhi >> 8), 1);  // (r >> 40) of the next position
        const float xf = (float)x;
        const float lhs = ff.x * ((float)nhi * 5.9604644775390625e-08f) + ff.y;
        const float ef = __expf(-0.5f * xf * xf);
        uint64_t accmask = __ballot(lhs < ef);
        const uint64_t zeromask = __ballot(idx == 0);
        if (__ballot(fabsf(lhs - ef) <= 1e-5f) & m) {
          if (!have64) {
            uint32_t D2[4] = {Dv[0], Dv[1], Dv[2], Dv[3]};
            asm volatile("" : "+v"(D2[0]), "+v"(D2[1]), "+v"(D2[2]), "+v"(D2[3]));  // (a front of its own: not to be merged with the round's)
            const FrontOut f2 = itemgen_front(Gl, S0v, D2);
            const uint64_t r2 = ((uint64_t)f2.rhi << 32) | f2.rlo;
            const int idx2 = (int)(r2 & 0xff);
            const uint64_t rabs2 = (r2 >> 9) & 0x000fffffffffffffull;
            const double rd2 = __longlong_as_double((long long)(rabs2 | 0x4330000000000000ull)) - 4503599627370496.0;
            double x2 = rd2 * __longlong_as_double((long long)z->kw[idx2].y);
            x2 = __longlong_as_double(__double_as_longlong(x2) ^ (long long)((r2 & 0x100ull) << 55));
            const int ic = idx2 > 0 ? idx2 : 1;
            const uint32_t nlo = (uint32_t)__shfl_down((int)(uint32_t)(r2 >> 11), 1),
                           nh2 = (uint32_t)__shfl_down((int)(uint32_t)(r2 >> 43), 1);
            const double u1 = (double)(((uint64_t)nh2 << 32) | nlo) * (1.0 / 9007199254740992.0);
            const double f1 = __longlong_as_double((long long)a.zfi[ic - 1]), f0 = __longlong_as_double((long long)a.zfi[ic]);
            acc64 = __ballot(itemgen_wedge_f64(x2, u1, f1, f0));
            have64 = true;
            continue;
          }
          accmask = acc64;
          have64 = false;
        }
        do {
          const int f = (int)__ffsll((long long)m) - 1;
          if ((zeromask >> f) & 1ull) {
            tailf = f;
            endpos = f;
            break;
          }
          if (f == 62) {
            endpos = 62;
            break;
          }
          const uint64_t acc = (accmask >> f) & 1ull;
          dead |= (2ull | (acc ^ 1ull)) << f;
          const int keep = f + 2;
          const uint64_t low = umask & ((1ull << keep) - 1ull);
          umask = low | ((umask >> (f + (int)acc)) << keep);
          m = missmask & ~umask & (~0ull << keep);
        } while (m);
      }
      const int off = lane - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
      int total = endpos - (int)__popcll(dead);
      const int need = T - (int)W;
      const bool alive = !__builtin_amdgcn_inverse_ballot_w64(dead);
      if (total >= need) {
        const uint64_t b = __ballot(alive && off == need - 1);
        endpos = (int)__ffsll((long long)b);
        total = need;
        tailf = -1;
      }
      if (alive && off < total) {
        // a step's uniform travels as its 53 random bits (r >> 11)
        const uint64_t ubits = ((uint64_t)(rhi >> 11) << 32) | __builtin_amdgcn_alignbit(rhi, rlo, 11);
        out[W + off] = __builtin_amdgcn_inverse_ballot_w64(umask) ? __longlong_as_double((long long)ubits) : x;
      }
      if (tailf >= 0) {
        const uint64_t sthi = ((uint64_t)rl32(fr.st[3], tailf) << 32) | rl32(fr.st[2], tailf),
                       stlo = ((uint64_t)rl32(fr.st[1], tailf) << 32) | rl32(fr.st[0], tailf);
        const ItemTail tl = itemgen_tail(sthi, stlo, inc.hi, inc.lo, (rl32(rabs_lo, tailf) >> 8) & 1u);  // numpy: (rabs >> 8) & 1
        if (lane == 0) out[W + total] = tl.xf;
        ++total;
        S0v[0] = sfirst((uint32_t)tl.s0lo);
        S0v[1] = sfirst((uint32_t)(tl.s0lo >> 32));
        S0v[2] = sfirst((uint32_t)tl.s0hi);
        S0v[3] = sfirst((uint32_t)(tl.s0hi >> 32));
        Dv[0] = sfirst((uint32_t)tl.dlo);
        Dv[1] = sfirst((uint32_t)(tl.dlo >> 32));
        Dv[2] = sfirst((uint32_t)tl.dhi);
        Dv[3] = sfirst((uint32_t)(tl.dhi >> 32));
      } else {
        // positions [0, endpos) are consumed (1 <= endpos <= 63): the states of lanes endpos - 1 and endpos are the
        // walker's new S_0 and S_1 (v_readlane: an SGPR each); d = S_1 - S_0 by a scalar borrow chain
        uint32_t S1v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          S0v[q] = rl32(fr.st[q], endpos - 1);
          S1v[q] = rl32(fr.st[q], endpos);
        }
        sub128_limbs(S1v, S0v, Dv);
      }
      W += (uint32_t)total;
    }
    if (lane == 0 && a.rng_out) {
      uint64_t* o = a.rng_out + (size_t)w * 4;
      o[0] = ((uint64_t)S0v[3] << 32) | S0v[2];
      o[1] = ((uint64_t)S0v[1] << 32) | S0v[0];
      o[2] = inc.hi;
      o[3] = inc.lo;
    }
  }
}

// ---- the throughput generator as a pass of its own (round 5) ---------------------------------------------------
// Philox4x32-10 is counter based: word p of walker w's stream is a function of (seed, subsequence = seq0 + w, p)
// alone, so the stream the walk consumes -- per step nb = ceil(n / 4) blocks of four Box-Muller normals and one
// uniform double, ph_stride = 4 nb + 2 words, exactly as walk.hip's lane kernel draws them from hiprand's state --
// needs no sequential pass and no generator state at all.  Inside the walk kernel (RNGQ_PHILOX: hiprand_init +
// hiprand_normal4 per block and step, an LDS ring, two wave_syncs per step, in a 187-register kernel at two
// wavefronts per SIMD) it cost as much as the walk itself.  Here one thread computes ONE Philox block (ten rounds,
// two v_mad_u64_u32 each) at full occupancy: the nb + 2 threads of a (walker, step) hold the blocks the step's words
// come from, a thread takes the words that spill over from its neighbour's block (the step's first word sits at
// position (offset + step ph_stride) mod 4 of a block), applies rocrand's own transforms (normal_distribution4,
// uniform_distribution_double: what hiprand_normal4 / hiprand_uniform_double apply to these words) and stores its 16
// bytes: a step's row is 4 (nb + 1) floats -- the normals as the fp32 values they are, the uniform as a double in
// the last slot -- 128 bytes at n = 25.  rwalkq_kernel<.., RNGQ_ITEMS32> reads a step's row one step ahead.
// Same words, same transforms: bit for bit the RNGQ_PHILOX kernel's and the lane kernel's walkers
// (tests/test_gpu_rwalkq.py::test_quad_philox_equals_lane_philox).
struct PhiloxItemArgs {
  float* items;
  unsigned long long seed, seq0, offset;
  const int* run_mode;
  int k, n, walks, wpr, my_mode, wbase;
  int gmagic;  // floor(1024 / (nb + 2)) + 1: lane / (nb + 2) = (lane * gmagic) >> 10 for lane < 64
  // presort workgroups (dh_ctx::PresortReq): the first ps_runs workgroups of the grid (PRESORT form only)
  const double* ps_keys;
  unsigned short* ps_out;
  int ps_n, ps_runs, ps_stride;
};

__device__ __forceinline__ uint4 philox_round(uint4 c, uint32_t k0, uint32_t k1) {
  const uint64_t m0 = mad_u64_u32(0xD2511F53u, c.x, 0ull), m1 = mad_u64_u32(0xCD9E8D57u, c.z, 0ull);
  return make_uint4((uint32_t)(m1 >> 32) ^ c.y ^ k0, (uint32_t)m1, (uint32_t)(m0 >> 32) ^ c.w ^ k1, (uint32_t)m0);
}
__device__ __forceinline__ uint4 philox_block(uint4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    c = philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

template <bool PRESORT>
// (waves_per_eu: the presort's sort is a call shared with itemgen_kernel<true>; its registers are bounded by the tightest caller)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) philox_items_kernel(PhiloxItemArgs a) {
  int block = (int)blockIdx.x;
  if constexpr (PRESORT) {
    __shared__ __attribute__((aligned(16))) unsigned char ps_smem[kPresortLds];
    if (block < a.ps_runs) {
      itemgen_presort(a.ps_keys + (size_t)block * a.ps_n, a.ps_out + (size_t)block * a.ps_stride, a.ps_n, ps_smem);
      return;
    }
    block -= a.ps_runs;
  }
  const int nb = (a.n + 3) >> 2, G = nb + 2, gpw = 64 / G, rs = 4 * (nb + 1);
  const int lane = threadIdx.x & 63;
  // (walker, step) of the wavefront's first group by scalar arithmetic, of this lane's group by a few adds: no
  // per-lane division (k walks < 2^31: the launcher's chunks see to it)
  const int wv = (int)sfirst(block * 4 + (threadIdx.x >> 6));
  const int gi = (lane * a.gmagic) >> 10, q = lane - gi * G;  // lane / G for lane < 64 (host-checked magic)
  const int total = a.k * a.walks;
  const int ws0 = wv * gpw;
  int w = ws0 / a.walks, step = ws0 - w * a.walks + gi;
  while (step >= a.walks) {
    step -= a.walks;
    ++w;
  }
  int ws = ws0 + gi;
  bool act = gi < gpw && ws < total;
  if (ws >= total) {
    ws = total - 1;
    w = a.k - 1;
    step = a.walks - 1;
  }
  if (a.run_mode && a.run_mode[(a.wbase + w) / a.wpr] != a.my_mode) act = false;
  if (!__any(act)) return;
  const unsigned long long p0 = a.offset + (unsigned long long)step * (unsigned long long)(4 * nb + 2);
  const int sub = (int)(p0 & 3ull);
  const unsigned long long Q = (p0 >> 2) + (unsigned long long)q, seq = a.seq0 + (unsigned long long)(a.wbase + w);
  // rocrand's restart(subsequence, offset): counter = (block index, subsequence) as two 64-bit halves
  const uint4 R = philox_block(make_uint4((uint32_t)Q, (uint32_t)(Q >> 32), (uint32_t)seq, (uint32_t)(seq >> 32)),
                               (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  uint4 Rn;
  Rn.x = (uint32_t)__shfl_down((int)R.x, 1);
  Rn.y = (uint32_t)__shfl_down((int)R.y, 1);
  Rn.z = (uint32_t)__shfl_down((int)R.z, 1);
  const uint4 v = sub == 0   ? R
                  : sub == 1 ? make_uint4(R.y, R.z, R.w, Rn.x)
                  : sub == 2 ? make_uint4(R.z, R.w, Rn.x, Rn.y)
                             : make_uint4(R.w, Rn.x, Rn.y, Rn.z);
  if (!act || q > nb) return;
  float* row = a.items + (size_t)ws * rs;
  if (q < nb) {
    *reinterpret_cast<float4*>(row + 4 * q) = rocrand_device::detail::normal_distribution4(v);
  } else {
    *reinterpret_cast<double*>(row + 4 * nb) = rocrand_device::detail::uniform_distribution_double(v.x, v.y);
  }
}

// fragments of a D x D row-major matrix M for the MFMA A operand: F[mt][s] = M[16 mt + (lane & 15)][4 s + (lane >> 4)]
template <int NR, int MT>
__device__ __forceinline__ void load_frags(const double* M, int n, int j, int t, double (&F)[MT][NR]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < NR; ++s) {
      const int row = 16 * mt + j, col = 4 * s + t;
      F[mt][s] = (row < n && col < n) ? M[row * n + col] : 0.0;
    }
}

// KS: the K steps taken by matrix instructions (NR, or NR - 1 in the R1 form of rwalkq_kernel, which adds the last
// column by vector instructions)
template <int NR, int MT, int KS = NR>
__device__ __forceinline__ void frag_matvec(const double (&F)[MT][NR], const double (&x)[NR], mfma_acc (&acc)[MT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = DH_MFMA_F64(F[mt][s], x[s], acc[mt]);
}

// log-likelihood of the walker's v (quarter vector per sub-lane); every sub-lane returns the same bits
// R1 (n = 4 (NR - 1) + 1): the last K step holds ONE live column; it is added as w[row] += P[row][n - 1] v[n - 1] by
// vector instructions (Cp: this lane's rows of that column) instead of a whole matrix instruction per row block
template <int NR, int MT, int KIND, bool R1 = false>
__device__ __forceinline__ double loglike_quad(const ProblemDev& P, int n, int t, const double (&v)[NR],
                                               const double* sprec, int lane, const double (&Cp)[NR]) {
  cdptr lp = as_const(P.like_par);
  const int lid = like_of<KIND>(P);
  if (lid == LIKE_GAUSS_PREC) {
    // q = v . (P v): P v of the wave's 16 walkers as one matrix product, fragments of P from LDS
    mfma_acc w[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) w[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < (R1 ? NR - 1 : NR); ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) w[mt] = DH_MFMA_F64(sprec[(mt * NR + s) * 64 + lane], v[s], w[mt]);
    if constexpr (R1) {
      const double vb = __shfl(v[NR - 1], lane & 15);  // element n - 1 of the walker: sub-lane 0's last register
#pragma unroll
      for (int r = 0; r < NR; ++r) w[r >> 2][r & 3] = fma(Cp[r], vb, w[r >> 2][r & 3]);
    }
    double q0 = 0.0, q1 = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r & 1)
        q1 = fma(v[r], w[r >> 2][r & 3], q1);
      else
        q0 = fma(v[r], w[r >> 2][r & 3], q0);
    }
    return lp[0] - 0.5 * grp_sum(q0 + q1);
  } else if (lid == LIKE_EGGBOX) {
    const double tmax = lp[0];
    double prod = 1.0;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (r < NR - 1 || 4 * r + t < n) prod *= cos((2.0 * tmax * v[r] - tmax) / 2.0);
    const double b = 2.0 + grp_prod(prod);
    const double b2 = b * b;
    return b2 * b2 * b;
  } else {
    double q0 = 0.0, q1 = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {  // padded entries are 0
      if (r & 1)
        q1 = fma(v[r], v[r], q1);
      else
        q0 = fma(v[r], v[r], q0);
    }
    return lp[0] - 0.5 * grp_sum(q0 + q1);
  }
}

template <int NR, int KIND>
__device__ __forceinline__ void prior_quad(const ProblemDev& P, int n, int t, const double (&u)[NR], double (&v)[NR]) {
  // (PRIOR_NORMAL: ndtri behind a call -- inlined, ocml's erfcinv next to the resident fragments spilled ~160
  // VGPRs and kept such problems on the lane-per-walker kernel until round 4)
  const int pid = prior_of<KIND>(P);
  if (pid == PRIOR_NORMAL) {
    cdptr pp = as_const(P.prior_par);
    const double mu = pp[0], sg = pp[1];
#pragma unroll 1
    for (int r = 0; r < NR; ++r) {
      double x = 0.5;
#pragma unroll
      for (int q = 0; q < NR; ++q) x = q == r ? u[q] : x;
      const double o = (r < NR - 1 || 4 * r + t < n) ? mu + sg * ndtri_far(x) : 0.0;
#pragma unroll
      for (int q = 0; q < NR; ++q) v[q] = q == r ? o : v[q];
    }
  } else if (pid == PRIOR_AFFINE) {
    cdptr pp = as_const(P.prior_par);
    const double a = pp[0], b = pp[1];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (r < NR - 1 || 4 * r + t < n) ? a * (2.0 * u[r] - 1.0) + b : 0.0;
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (r < NR - 1 || 4 * r + t < n) ? u[r] : 0.0;
  }
}

// ur^(1/n) for ur in [0, 1), n = 4 (NR - 1) + k with k in 1..4 (bounding.py:1295: the radius of a point uniform in
// the n-ball).  ocml's log + exp are ~90 instructions that all four lanes of a walker repeat every step; here a
// single-precision seed (relative error ~1e-7) and two Newton steps on y^n = ur, y^n by a multiplication chain
// that is fixed at compile time up to the factor y^k (wave-uniform selects).  The second step's correction is
// ~1e-13, so the quotient only needs the seed's precision; the result is good to ~2e-16.
template <int NR>
__device__ __forceinline__ double pow_n(double y, int k) {
  const double y2 = y * y, y4 = y2 * y2;
  double z = 1.0, b = y4;  // z = y4^(NR - 1)
#pragma unroll
  for (int e = NR - 1; e > 0; e >>= 1) {
    if (e & 1) z = (z == 1.0 && e == NR - 1) ? b : z * b;
    if (e > 1) b *= b;
  }
  const double yk = k == 1 ? y : k == 2 ? y2 : k == 3 ? y * y2 : y4;
  return z * yk;
}
template <int NR>
__device__ __forceinline__ double root_n(double ur, int n, double inv_n) {
  const int k = n - 4 * (NR - 1);
  double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)ur) * (float)inv_n);
  double p = pow_n<NR>(y, k);
  const double rp = (double)__builtin_amdgcn_rcpf((float)p);
  y = fma(-(y * ((p - ur) * rp)), inv_n, y);
  p = pow_n<NR>(y, k);
  y = fma(-(y * ((p - ur) * rp)), inv_n, y);
  return ur > 0.0 ? y : 0.0;
}

// generic_random_walk (internal_samplers.py:866-986) for ndim == ncdim: four lanes per walker.  Workgroup = 4 wavefronts = 64 walkers.
// R1 (round 6): n = 4 (NR - 1) + 1 -- the headline's D = 25 -- leaves ONE live column in the last K step of both
// products; a matrix instruction per row block (2 x 86 cycles of the pipe) for it is replaced by NR multiply-adds and a
// lane broadcast.  A matrix instruction's K steps are fused multiply-adds onto the accumulator and the padded products
// are exact zeros, so fma(column, x, acc) is what the last instruction computed: the same bits
// (tests/test_gpu_rwalkq.py::test_last_column_by_vector_instructions_equals_the_matrix_form).
template <int NR, int KIND, int RNG, bool R1 = false>
__global__ void __launch_bounds__(256) rwalkq_kernel(RwalkQArgs a) {
  constexpr int MT = (4 * NR + 15) / 16;
  constexpr int STRIDE = RingGeom<NR>::stride;
  __shared__ __attribute__((aligned(16))) char zig_mem[RNG == RNGQ_PCG64 ? sizeof(ZigQ) : 16];
  __shared__ double ring_all[(RNG == RNGQ_ITEMS || RNG == RNGQ_ITEMS32) ? 1 : 64 * STRIDE];  // [walker slot][ring position]: the walkers' next items
  __shared__ double sprec[MT * NR * 64];       // MFMA fragments of the precision matrix
  __shared__ WaveGenLds gen_all[RNG == RNGQ_PCG64 ? 4 : 1];
  ZigQ& zig = *reinterpret_cast<ZigQ*>(zig_mem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane >> 4, j = lane & 15;
  const int slot = wave * 16 + j;
  const int w = blockIdx.x * 64 + slot;
  const bool live = w < a.k;
  const int wi = live ? w : a.k - 1;  // dead lanes shadow the last walker (no stores)
  const int n = a.ndim;
  if constexpr (RNG == RNGQ_PCG64) {
    for (int i = tid; i < 256; i += 256) {
      zig.kw[i] = make_ulonglong2(a.zki[i], a.zwi[i]);
      const int im = i > 0 ? i - 1 : 0;
      const double fa = __longlong_as_double((long long)a.zfi[im]), fb = __longlong_as_double((long long)a.zfi[i]);
      zig.ff[i] = make_float2((float)(fa - fb), (float)fb);
    }
  }
  if (like_of<KIND>(a.prob) == LIKE_GAUSS_PREC) {
    const double* Pm = a.prob.like_par + 1;
    for (int f = tid; f < MT * NR * 64; f += 256) {
      const int l = f & 63, s = (f >> 6) % NR, mt = (f >> 6) / NR;
      const int row = 16 * mt + (l & 15), col = 4 * s + (l >> 4);
      sprec[f] = (row < n && col < n) ? Pm[row * n + col] : 0.0;
    }
  }
  __syncthreads();
  // from here on the wavefronts are on their own (no workgroup barrier below)
  double loglstar = a.loglstar, scale = a.scale;
  bool on = true;
  if (a.run_mode) {
    const int run = (a.wbase + wi) / a.wpr;
    on = a.run_mode[run] == a.my_mode;
    loglstar = a.run_loglstar[run];
    scale = a.run_scale[run];
  }
  // a wavefront leaves only as a whole: the matrix instructions take operands from all 64 lanes
  if (!__any(on)) return;

  double u[NR], up[NR], dr[NR], vv[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) u[r] = (r < NR - 1 || 4 * r + t < n) ? a.u0[(size_t)wi * n + 4 * r + t] : 0.5;
  WaveGenLds* gen = &gen_all[wave];
  Limbs128 Gl = {{0u, 0u, 0u, 0u}};
  hiprandStatePhilox4_32_10_t ph;
  const int n1 = n + 1, T = a.walks * n1;
  WaveGenConst gk;
  gk.n = n;
  gk.n1 = n1;
  gk.T = T;
  gk.cap = ring_cap(n1);
  gk.magic_n1 = (uint32_t)(0x100000000ull / (uint32_t)n1) + 1u;
  gk.magic_cap = (uint32_t)(0x100000000ull / (uint32_t)gk.cap) + 1u;
  gk.zfi = a.zfi;
  gk.U0 = 0;
  for (int b = 0; b < 64; b += n1) gk.U0 |= 1ull << b;
  double* ring = ring_all + wave * 16 * STRIDE;  // this wavefront's sixteen rows
  double* myrow = ring + j * STRIDE;
  const int walker0 = blockIdx.x * 64 + wave * 16;
  if constexpr (RNG == RNGQ_PCG64) {
    if (lane < 16) wavegen_init(gen, lane, a.rng_in + (size_t)wi * 4);
    wave_sync();
    const U128 gj = {a.pcg_jump[2 * lane], a.pcg_jump[2 * lane + 1]};  // G_{lane + 1}
    Gl = opaque_limbs(gj);
  }
  const bool lastok = 4 * (NR - 1) + t < n;
  // boundary conditions of this lane's coordinates, two bits each (padding: hard, and it sits at 0.5)
  uint32_t bcp = 0;
  if (a.bc) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (r < NR - 1 || lastok) bcp |= ((uint32_t)a.bc[4 * r + t] & 3u) << (2 * r);
  }
  // RNGQ_ITEMS: the walker's stream in global memory and the registers that hold the coming step's items
  const double* myitems = a.items + (size_t)wi * T;
  double nx[NR], nxu = 0.0;
  if constexpr (RNG == RNGQ_ITEMS) {
#pragma unroll
    for (int r = 0; r < NR; ++r) nx[r] = myitems[(r < NR - 1 || lastok) ? 4 * r + t : 0];
    nxu = myitems[n];
  }
  // RNGQ_ITEMS32: rows of 4 (nb + 1) floats per step (philox_items_kernel)
  const int rs32 = 4 * (((n + 3) >> 2) + 1);
  const float* myitems32 = a.items32 + (size_t)wi * a.walks * rs32;
  float nxf[NR];
  if constexpr (RNG == RNGQ_ITEMS32) {
#pragma unroll
    for (int r = 0; r < NR; ++r) nxf[r] = myitems32[4 * r + t];  // (NR = nb: a row holds 4 nb normals)
    nxu = *reinterpret_cast<const double*>(myitems32 + 4 * NR);
  }
  int start = 0;  // ring position of the current step's first item: (step * n1) mod cap
  WQP(WqProf pf; pf.fill = pf.rest = pf.rounds = pf.segs = pf.wedges = pf.t0 = 0;)
  const int nb = (n + 3) >> 2;           // hiprand_normal4 blocks per step
  const int ph_stride = 4 * nb + 2;      // 32-bit draws per step of the lane-per-walker Philox kernel

  const int my_frame = a.axes_idx ? a.axes_idx[wi] : 0;
  const int f0 = __builtin_amdgcn_readfirstlane(my_frame);
  const bool uni = __all(my_frame == f0);
  double F[MT][NR];
  load_frags<NR, MT>(a.axes + (size_t)f0 * n * n, n, j, t, F);
  // R1: this lane's rows 4 r + t of the last column of the frame and of the precision matrix
  double Cf[NR], Cp[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    Cf[r] = Cp[r] = 0.0;
    if constexpr (R1) {
      const int row = 4 * r + t;
      if (row < n) {
        Cf[r] = a.axes[(size_t)f0 * n * n + (size_t)row * n + (n - 1)];
        if (like_of<KIND>(a.prob) == LIKE_GAUSS_PREC) Cp[r] = a.prob.like_par[1 + (size_t)row * n + (n - 1)];
      }
    }
  }

  int nacc = 0, nrej = 0;
  double logl_cur = 0.0;
  const double inv_n = 1.0 / (double)n;

#pragma unroll 1
  for (int step = 0; step < a.walks; ++step) {
    // randsphere (bounding.py:1288-1297): n normals, one uniform
    if constexpr (RNG == RNGQ_ITEMS || RNG == RNGQ_ITEMS32) {
      // this step's items were loaded a step ahead (below)
    } else if constexpr (RNG == RNGQ_PCG64) {
      WQP(const long long tq0 = clock64();)
      wavegen_fill<STRIDE>(gen, (step + 1) * n1, Gl, &zig, ring, lane, gk, a.rng_in, walker0, a.k WQP(, &pf));
      WQP(const long long tq1 = clock64(); pf.fill += tq1 - tq0; if (step) pf.rest += tq0 - pf.t0; pf.t0 = tq1;)
    } else {
      // the walker's Philox subsequence exactly as walk.hip consumes it (per step: nb blocks of four
      // normals, one uniform double), block b drawn by sub-lane b & 3
      for (int b = t; b <= nb; b += 4) {
        hiprand_init(a.ph_seed, a.ph_seq0 + (unsigned long long)wi,
                     a.ph_offset + (unsigned long long)step * ph_stride + 4ull * b, &ph);
        if (b < nb) {
          const float4 zf = hiprand_normal4(&ph);
          const double zz[4] = {(double)zf.x, (double)zf.y, (double)zf.z, (double)zf.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (4 * b + i < n) myrow[start + 4 * b + i] = zz[i];
        } else {
          myrow[start + n] = hiprand_uniform_double(&ph);
        }
      }
    }
    double ss = 0.0, ur;
    if constexpr (RNG == RNGQ_ITEMS) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        dr[r] = (r < NR - 1 || lastok) ? nx[r] : 0.0;
        ss = fma(dr[r], dr[r], ss);
      }
      ur = (double)(uint64_t)__double_as_longlong(nxu) * (1.0 / 9007199254740992.0);
      // the next step's items: in flight while this step computes (the last step reads its own again)
      const double* nrow = myitems + (step + 1 < a.walks ? (step + 1) * n1 : step * n1);
#pragma unroll
      for (int r = 0; r < NR; ++r) nx[r] = nrow[(r < NR - 1 || lastok) ? 4 * r + t : 0];
      nxu = nrow[n];
    } else if constexpr (RNG == RNGQ_ITEMS32) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        dr[r] = (r < NR - 1 || lastok) ? (double)nxf[r] : 0.0;
        ss = fma(dr[r], dr[r], ss);
      }
      ur = nxu;  // hiprand_uniform_double's value, in (0, 1]
      const float* nrow = myitems32 + (size_t)(step + 1 < a.walks ? step + 1 : step) * rs32;
#pragma unroll
      for (int r = 0; r < NR; ++r) nxf[r] = nrow[4 * r + t];
      nxu = *reinterpret_cast<const double*>(nrow + 4 * NR);
    } else {
      wave_sync();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        // a step's items lie in a row without wrapping; NR = ceil(n / 4): only the last register of a vector has
        // lanes past the dimension
        const double item = myrow[start + 4 * r + t];
        dr[r] = (r < NR - 1 || lastok) ? item : 0.0;
        ss = fma(dr[r], dr[r], ss);
      }
      ur = myrow[start + n];
      if constexpr (RNG == RNGQ_PCG64)  // the generator hands over the uniform's 53 random bits
        ur = (double)(uint64_t)__double_as_longlong(ur) * (1.0 / 9007199254740992.0);
      wave_sync();
      start += n1;
      start = start >= gk.cap ? 0 : start;
    }
    ss = grp_sum(ss);
    // scale * ur^(1/n) / |dr| (bounding.py:1295-1296).  This is per-walker scalar work that all four
    // sub-lanes repeat, so it is kept short: exp(log(ur) / n) instead of ocml's double-double pow (|log
    // ur| / n is O(1): the product costs no accuracy), 1 / sqrt by v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(ss);
    y = y * fma(-0.5 * ss * y, y, 1.5);
    y = y * fma(-0.5 * ss * y, y, 1.5);
    const double fac = scale * (root_n<NR>(ur, n, inv_n) * y);
    // du = axes @ dr on the matrix cores; walkers of a wave on different frames: one product per frame
    mfma_acc acc[MT];
    if (uni) {
      frag_matvec<NR, MT, R1 ? NR - 1 : NR>(F, dr, acc);
      if constexpr (R1) {
        const double xb = __shfl(dr[NR - 1], j);
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r >> 2][r & 3] = fma(Cf[r], xb, acc[r >> 2][r & 3]);
      }
    } else {
      bool done = false;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
      for (;;) {
        const uint64_t rem = __ballot(!done);
        if (!rem) break;
        const int cur = __shfl(my_frame, __ffsll((long long)rem) - 1);
        // (fragments of its own: loaded into F, the wavefronts of ONE frame -- the common case -- paid 32 register
        // copies per step for keeping their F intact, round 6)
        double Fn[MT][NR];
        load_frags<NR, MT>(a.axes + (size_t)cur * n * n, n, j, t, Fn);
        mfma_acc tmp[MT];
        frag_matvec<NR, MT>(Fn, dr, tmp);
        if (my_frame == cur) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = tmp[mt];
          done = true;
        }
      }
    }
    double lo = 0.5, hi = 0.5;
#pragma unroll
    for (int r = 0; r < NR; ++r) up[r] = (r < NR - 1 || lastok) ? fma(fac, acc[r >> 2][r & 3], u[r]) : 0.5;
    bool inside_q;
    if (a.bc) {
      // periodic wrap / reflection, then unitcheck with the wider interval on those coordinates
      // (utils.py:1036-1078; the same rule as the lane-per-walker kernel)
      inside_q = true;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const uint32_t b = (bcp >> (2 * r)) & 3u;
        double x = up[r];
        if (b == DH_BC_PERIODIC) x = wrap01(x);
        if (b == DH_BC_REFLECT) x = reflect01(x);
        up[r] = x;
        inside_q = inside_q && (b == DH_BC_HARD ? (x > 0.0) && (x < 1.0) : (x > -0.5) && (x < 1.5));
      }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        lo = fmin(lo, up[r]);
        hi = fmax(hi, up[r]);
      }
      // unitcheck (utils.py:1036-1050) over the four quarters of the walker
      inside_q = (lo > 0.0) && (hi < 1.0);
    }
    const bool inside = grp_bits(__ballot(inside_q), j) == 0x0001000100010001ull;
    // a proposal outside the cube is counted as a call and a reject, no likelihood evaluated; here the
    // evaluation runs anyway (the matrix instruction serves the whole wave) and its verdict is ignored
    prior_quad<NR, KIND>(a.prob, n, t, up, vv);
    const double ll = loglike_quad<NR, MT, KIND, R1>(a.prob, n, t, vv, sprec, lane, Cp);
    if (inside && ll > loglstar) {
#pragma unroll
      for (int r = 0; r < NR; ++r) u[r] = up[r];
      logl_cur = ll;
      ++nacc;
    } else {
      ++nrej;
    }
  }
  // v of the returned point; logl is re-evaluated when nothing was accepted (internal_samplers.py:970-975)
  prior_quad<NR, KIND>(a.prob, n, t, u, vv);
  const double ll0 = loglike_quad<NR, MT, KIND, R1>(a.prob, n, t, vv, sprec, lane, Cp);
  if (nacc == 0) logl_cur = ll0;
  if (live && on) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (r < NR - 1 || lastok) {
        a.u[(size_t)w * n + 4 * r + t] = u[r];
        a.v[(size_t)w * n + 4 * r + t] = vv[r];
      }
#ifdef DH_WQ_PROF
    if (lane == 0) {
      double* o = a.v + (size_t)w * n;
      o[0] = (double)pf.fill; o[1] = (double)pf.rest; o[2] = (double)pf.rounds; o[3] = (double)pf.segs;
      o[4] = (double)pf.wedges;
    }
#endif
    if (t == 0) {
      a.logl[w] = logl_cur;
      a.nacc[w] = nacc;
      a.nrej[w] = nrej;
      if (RNG == RNGQ_PCG64 && a.rng_out) {
        // every item of the walk is drawn: S_0 is the generator state after the last of them
        uint64_t* o = a.rng_out + (size_t)w * 4;
        const uint32_t* gs = gen->row[j].s;
        o[0] = ((uint64_t)gs[3] << 32) | gs[2];
        o[1] = ((uint64_t)gs[1] << 32) | gs[0];
        o[2] = a.rng_in[(size_t)w * 4 + 2];
        o[3] = a.rng_in[(size_t)w * 4 + 3];
      }
    }
  }
}

}  // namespace

namespace dh {

// Eligible launches (rwalk_launch_runs decides): ndim == ncdim in 2..32, fused likelihood and prior.  Returns DH_OK after enqueueing on the context's stream.
int rwalkq_launch(dh_ctx* ctx, const ProblemDev& prob, int k, int ndim, const double* u0, const double* axes, int m,
                  const int32_t* axes_idx, double scale, double loglstar, int walks, const uint64_t* rng, double* u,
                  double* v, double* logl, int32_t* naccept, int32_t* nreject, uint64_t* rng_out,
                  const double* run_loglstar, const double* run_scale, const int* run_mode, int wpr, int my_mode,
                  const PhiloxKey* philox, const int8_t* bc) {
  RwalkQArgs a;
  a.prob = prob;
  a.bc = bc;
  a.k = k;
  a.ndim = ndim;
  a.walks = walks;
  a.m = m;
  a.scale = scale;
  a.loglstar = loglstar;
  a.u0 = u0;
  a.axes = axes;
  a.axes_idx = axes_idx;
  a.rng_in = rng;
  a.u = u;
  a.v = v;
  a.logl = logl;
  a.nacc = naccept;
  a.nrej = nreject;
  a.rng_out = rng_out;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  a.pcg_jump = ctx->pcg_jump();
  a.run_loglstar = run_loglstar;
  a.run_scale = run_scale;
  a.run_mode = run_mode;
  a.wpr = wpr;
  a.my_mode = my_mode;
  a.ph_seed = philox ? philox->seed : 0;
  a.ph_seq0 = philox ? philox->seq0 : 0;
  a.ph_offset = philox ? philox->offset : 0;
  a.items = nullptr;
  a.items32 = nullptr;
  a.wbase = 0;
  const dim3 block(256);
  const int kind = problem_kind(prob.like_id, prob.prior_id) == KIND_PREC_AFFINE ? KIND_PREC_AFFINE : KIND_GENERIC;
  const int nr = (ndim + 3) / 4;  // 2 <= ndim <= 32: 1 .. 8
  // the last column by vector instructions where it is the only live one of its K step (DH_RWALKQ_R1=0: the matrix form)
  const bool r1 = ndim % 4 == 1 && ndim >= 5 && !(getenv("DH_RWALKQ_R1") && atoi(getenv("DH_RWALKQ_R1")) == 0);
#define L(NRR, KK, GRID)                                                                               \
  do {                                                                                                 \
    constexpr bool R1K = KK == KIND_PREC_AFFINE && NRR >= 2;                                            \
    if (a.items32 && r1 && R1K)                                                                        \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_ITEMS32, R1K>), GRID, block, 0, ctx->stream, a); \
    else if (a.items32)                                                                                \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_ITEMS32>), GRID, block, 0, ctx->stream, a);      \
    else if (philox)                                                                                   \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_PHILOX>), GRID, block, 0, ctx->stream, a);       \
    else if (a.items && r1 && R1K)                                                                     \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_ITEMS, R1K>), GRID, block, 0, ctx->stream, a);   \
    else if (a.items)                                                                                  \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_ITEMS>), GRID, block, 0, ctx->stream, a);        \
    else                                                                                               \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_PCG64>), GRID, block, 0, ctx->stream, a);        \
  } while (0)
#define X(NRR, GRID)                   \
  if (nr == NRR) {                     \
    if (kind == KIND_PREC_AFFINE)      \
      L(NRR, KIND_PREC_AFFINE, GRID);  \
    else                               \
      L(NRR, KIND_GENERIC, GRID);      \
  }
#define XALL(GRID) X(1, GRID) X(2, GRID) X(3, GRID) X(4, GRID) X(5, GRID) X(6, GRID) X(7, GRID) X(8, GRID)
  if (!ctx->rwalk_items) {
    const dim3 grid((k + 63) / 64);
    XALL(grid)
    return hip_ok(ctx, hipGetLastError(), "rwalkq launch") ? DH_OK : DH_ERR_HIP;
  }
  // The generator as a pass of its own (PCG64: itemgen_kernel, fp64 items; Philox: philox_items_kernel, fp32 rows):
  // walkers in chunks whose item streams fit the context's buffer
  const size_t per_walker = philox ? (size_t)walks * 4 * (nr + 1) * sizeof(float) : (size_t)walks * (ndim + 1) * sizeof(double);
  size_t chunk = ctx->items_budget / per_walker;
  if (chunk < 64) chunk = 64;
  if (chunk > (size_t)k) chunk = (size_t)k;
  chunk = (chunk + 63) / 64 * 64;
  const size_t need = (chunk < (size_t)k ? chunk : (size_t)k) * per_walker;
  if (need > ctx->items_cap) {
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    if (ctx->items) (void)hipFree(ctx->items);
    ctx->items = nullptr;
    ctx->items_cap = 0;
    if (!hip_ok(ctx, hipMalloc((void**)&ctx->items, need), "hipMalloc(rwalk item streams)")) return DH_ERR_NOMEM;
    ctx->items_cap = need;
  }
  for (size_t first = 0; first < (size_t)k; first += chunk) {
    const int kc = (int)((size_t)k - first < chunk ? (size_t)k - first : chunk);
    if (philox) {
      PhiloxItemArgs g;
      g.items = reinterpret_cast<float*>(ctx->items);
      g.seed = philox->seed;
      g.seq0 = philox->seq0;
      g.offset = philox->offset;
      g.run_mode = run_mode;
      g.k = kc;
      g.n = ndim;
      g.walks = walks;
      g.wpr = wpr;
      g.my_mode = my_mode;
      g.wbase = (int)first;
      g.gmagic = 1024 / (nr + 2) + 1;
      for (int l = 0; l < 64; ++l)
        if (((l * g.gmagic) >> 10) != l / (nr + 2)) return fail(ctx, DH_ERR_ARG, "rwalkq: lane-group magic");
      const long long gpw = 64 / (nr + 2), nws = (long long)kc * walks;
      if (nws >= (1ll << 31) - 64) return fail(ctx, DH_ERR_ARG, "rwalkq: %lld walker-steps in one generator launch", nws);
      const long long gwaves = (nws + gpw - 1) / gpw;
      g.ps_keys = nullptr;
      g.ps_out = nullptr;
      g.ps_n = g.ps_runs = g.ps_stride = 0;
      dh_ctx::PresortReq& pr = ctx->presort;  // (as in front of the PCG64 pass below)
      if (pr.keys && pr.runs > 0 && pr.n <= 2048 && first == 0 && kc == k && gwaves > 64ll * pr.runs) {
        g.ps_keys = pr.keys;
        g.ps_out = pr.out;
        g.ps_n = pr.n;
        g.ps_runs = pr.runs;
        g.ps_stride = pr.stride;
        pr.done = 1;
      }
      pr.keys = nullptr;
      if (g.ps_runs > 0)
        hipLaunchKernelGGL(philox_items_kernel<true>, dim3((unsigned)((gwaves + 3) / 4) + g.ps_runs), block, 0, ctx->stream, g);
      else
        hipLaunchKernelGGL(philox_items_kernel<false>, dim3((unsigned)((gwaves + 3) / 4)), block, 0, ctx->stream, g);
      a.items32 = reinterpret_cast<const float*>(ctx->items);
      a.ph_seq0 = philox->seq0 + first;
    } else {
      ItemGenArgs g;
      g.rng_in = rng + first * 4;
      g.rng_out = rng_out ? rng_out + first * 4 : nullptr;
      g.items = ctx->items;
      g.zki = ctx->zki();
      g.zwi = ctx->zwi();
      g.zfi = ctx->zfi();
      g.pcg_jump = ctx->pcg_jump();
      g.run_mode = run_mode;
      g.k = kc;
      g.n = ndim;
      g.walks = walks;
      g.wpr = wpr;
      g.my_mode = my_mode;
      g.wbase = (int)first;
      // one wavefront per walker up to the grid that is RESIDENT AT ONCE (the occupancy query's blocks per CU: six
      // at 74 registers); beyond that the wavefronts loop.  A grid of 8 blocks per CU (rounds 3-5a) ran a second
      // generation of blocks on a third of the wavefront slots: the same walkers per wavefront, a quarter of them at
      // a third of the occupancy (EXPERIMENTS.md R5.1)
      int gblocks = (kc + 3) / 4;
      int& per_cu = ctx->itemgen_blocks_per_cu;  // (per context: the library keeps no process-global state)
      if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, itemgen_kernel<false>, 256, 0) != hipSuccess || nb < 1) nb = 6;
        if (const char* e = getenv("DH_ITEMGEN_BLOCKS_PER_CU")) {
          const int v = atoi(e);
          if (v >= 1 && v <= 16) nb = v;
        }
        per_cu = nb;
      }
      const int gmax = ctx->num_cu * per_cu;
      if (gblocks > gmax) gblocks = gmax;
      // the resident loop's presort request rides in front of the generator's grid (the generator's wavefronts loop
      // over the walkers, so the `runs` slots it gives up cost it runs / grid of its time)
      g.ps_keys = nullptr;
      g.ps_out = nullptr;
      g.ps_n = g.ps_runs = g.ps_stride = 0;
      dh_ctx::PresortReq& pr = ctx->presort;
      if (pr.keys && pr.runs > 0 && pr.n <= 2048 && first == 0 && kc == k && gblocks > 4 * pr.runs) {
        g.ps_keys = pr.keys;
        g.ps_out = pr.out;
        g.ps_n = pr.n;
        g.ps_runs = pr.runs;
        g.ps_stride = pr.stride;
        pr.done = 1;
        if (gblocks + pr.runs <= gmax) gblocks += pr.runs;  // (room left: the generator keeps all its workgroups)
      }
      pr.keys = nullptr;
      if (g.ps_runs > 0)
        hipLaunchKernelGGL(itemgen_kernel<true>, dim3(gblocks), block, 0, ctx->stream, g);
      else
        hipLaunchKernelGGL(itemgen_kernel<false>, dim3(gblocks), block, 0, ctx->stream, g);
      a.items = ctx->items;
      a.rng_in = rng + first * 4;
    }
    a.k = kc;
    a.wbase = (int)first;
    a.u0 = u0 + first * ndim;
    a.axes_idx = axes_idx ? axes_idx + first : nullptr;
    a.u = u + first * ndim;
    a.v = v + first * ndim;
    a.logl = logl + first;
    a.nacc = naccept + first;
    a.nrej = nreject + first;
    a.rng_out = nullptr;  // written by the generator pass
    const dim3 grid((kc + 63) / 64);
    XALL(grid)
  }
#undef XALL
#undef X
#undef L
  return hip_ok(ctx, hipGetLastError(), "rwalkq launch") ? DH_OK : DH_ERR_HIP;
}

}  // namespace dh
