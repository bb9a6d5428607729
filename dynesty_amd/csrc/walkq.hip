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
#include "rng_pcg64.h"

using namespace dh;

namespace {

typedef double mfma_acc __attribute__((ext_vector_type(4)));
#define DH_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

enum : int { RNGQ_PCG64 = 0, RNGQ_PHILOX = 1 };

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
  const double* run_loglstar;
  const double* run_scale;
  const int* run_mode;
  int wpr, my_mode;
  unsigned long long ph_seed, ph_seq0, ph_offset;
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

// LCG jump constants of PCG64's multiplier: A_j = mult^j, G_j = 1 + mult + ... + mult^(j-1) (mod 2^128)
__device__ __forceinline__ U128 jump_A(int j) {  // j = 1..4
  const U128 a1 = {0x2360ed051fc65da4ull, 0x4385df649fccf645ull}, a2 = {0x17bce35bdf69743cull, 0x529ed9eb20e0ae99ull},
             a3 = {0x25f041404bd80e82ull, 0xeb5ae837ed42153dull}, a4 = {0xf4dd417327db7a9bull, 0xd194dfbe42d45771ull};
  return j == 1 ? a1 : j == 2 ? a2 : j == 3 ? a3 : a4;
}
__device__ __forceinline__ U128 jump_G(int j) {
  const U128 g1 = {0x0ull, 0x1ull}, g2 = {0x2360ed051fc65da4ull, 0x4385df649fccf646ull},
             g3 = {0x3b1dd060ff2fd1e0ull, 0x9624b94fc0ada4dfull}, g4 = {0x610e11a14b07e063ull, 0x817fa187adefba1cull};
  return j == 1 ? g1 : j == 2 ? g2 : j == 3 ? g3 : g4;
}
#define DH_PCG_MULT_INV_HI 0x07dda22b93979860ull  // mult^-1 mod 2^128
#define DH_PCG_MULT_INV_LO 0x98abc8b0716eac8dull

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
  double fi[256];
};

// One walker's PCG64 on four lanes.  S = the walker's generator state advanced t + 1 steps.
struct QuadPcg {
  U128 S, inc, AJ, TJ, T4;
  __device__ __forceinline__ void init(const uint64_t* p, int t) {
    const U128 base = {p[0], p[1]};
    inc.hi = p[2];
    inc.lo = p[3];
    AJ = jump_A(t + 1);
    TJ = mul128(jump_G(t + 1), inc);
    T4 = mul128(jump_G(4), inc);
    S = add128(mul128(base, AJ), TJ);
  }
  // the walker's generator state itself (meaningful on sub-lane 0: S = mult * base + inc)
  __device__ __forceinline__ U128 base() const {
    const U128 minv = {DH_PCG_MULT_INV_HI, DH_PCG_MULT_INV_LO};
    return mul128(sub128(S, inc), minv);
  }
};

// One step's draws of the reference (nc normals, then one uniform) -> items[i * 64 + slot], i = 0..nc.
// `slot` = the walker's column of the staging array; j = lane & 15, t = lane >> 4.
__device__ __forceinline__ void quad_draw_step(QuadPcg& q, const ZigQ* z, double* items, int slot, int j, int t,
                                               int nc) {
#pragma clang fp contract(off)
  const int NI = nc + 1;
  const U128 A4 = jump_A(4);
  int count = 0;      // items finished (the same in the four sub-lanes of a walker)
  bool pend = false;  // a candidate that missed the fast accept waits for its wedge uniform
  int pidx = 0;
  double px = 0.0;
  while (__any(count < NI)) {
    const bool act = count < NI;
    const uint64_t r = pcg_output(q.S);
    int shift = 0;
    if (__any(pend)) {
      // sub-lane 0's draw is the wedge uniform of the pending candidate (numpy distributions.c:
      // (fi[idx-1] - fi[idx]) * next_double() + fi[idx] < exp(-0.5 x x))
      const uint32_t rlo = grp32((uint32_t)r, j), rhi = grp32((uint32_t)(r >> 32), j);
      if (pend) {
        const double u1 = (double)((((uint64_t)rhi << 32) | rlo) >> 11) * (1.0 / 9007199254740992.0);
        if ((z->fi[pidx - 1] - z->fi[pidx]) * u1 + z->fi[pidx] < exp(-0.5 * px * px)) {
          if (t == 0) items[count * 64 + slot] = px;
          ++count;
        }
        shift = 1;
        pend = false;
      }
    }
    const int my = count + t - shift;
    const bool valid = act & (t >= shift) & (my < NI);
    const int idx = (int)(r & 0xff);
    const uint64_t rabs = (r >> 9) & 0x000fffffffffffffull;
    const double rd = __longlong_as_double((long long)(rabs | 0x4330000000000000ull)) - 4503599627370496.0;
    const ulonglong2 kw = z->kw[idx];  // looked up by every lane: no branch around the LDS read
    double x = rd * __longlong_as_double((long long)kw.y);
    x = __longlong_as_double(__double_as_longlong(x) ^ (long long)((r & 0x100ull) << 55));
    const bool isn = valid & (my < nc);
    const uint64_t kk = kw.x;
    const bool miss = isn & !(rabs < kk);
    const int fm = grp_first(grp_bits(__ballot(miss), j));       // first missing sub-lane of my walker
    int tend = shift + NI - count;                                // one past the last valid sub-lane
    tend = tend < 4 ? tend : 4;
    if (valid && t < fm) {
      double val = x;
      if (!isn) val = (double)(r >> 11) * (1.0 / 9007199254740992.0);  // the step's uniform: once in ~7 rounds
      items[my * 64 + slot] = val;
    }
    const int stop = fm < tend ? fm : tend;  // sub-lanes [shift, stop) were consumed as items
    if (act) count += stop - shift;
    // re-alignment: everyone's new state is A * B + T with (A, B, T) = (A4, own S, T4) when all four
    // candidates were consumed, else (A_{t+1}, S of the last consumed sub-lane, G_{t+1} inc)
    const bool hit = act && fm < tend;                     // a miss inside the valid range
    const bool rejump = act && (hit || tend < 4);
    U128 B = q.S, A = A4, T = q.T4;
    if (__any(rejump)) {
      const int src = hit ? fm : tend - 1;
      const int srclane = rejump ? ((src << 4) | j) : ((t << 4) | j);
      // every lane shuffles (no cross-lane reads under a divergent mask)
      const U128 Sf = grp128(q.S, srclane);
      const int fidx = (int)grp32((uint32_t)idx, srclane);
      const uint64_t xb = (uint64_t)__double_as_longlong(x);
      const double fx = __longlong_as_double(
          (long long)(((uint64_t)grp32((uint32_t)(xb >> 32), srclane) << 32) | grp32((uint32_t)xb, srclane)));
      if (hit && fidx == 0) {
        // tail of the distribution (idx == 0): finished sequentially, redundantly by the walker's four
        // lanes -- rare (about 3 in 10^4 draws)
        Pcg64 g;
        g.state = Sf;
        g.inc = q.inc;
        const uint64_t rf = pcg_output(Sf);
        const uint64_t rabsf = (rf >> 9) & 0x000fffffffffffffull;
        double xf;
        for (;;) {
          const double xx = -DH_ZIG_INV_R * log1p(-g.next_double());
          const double yy = -log1p(-g.next_double());
          if (yy + yy > xx * xx) {
            xf = ((rabsf >> 8) & 1) ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
            break;
          }
        }
        if (t == 0) items[count * 64 + slot] = xf;
        ++count;
        B = g.state;
      } else if (hit) {
        pend = true;
        pidx = fidx;
        px = fx;
        B = Sf;
      } else if (rejump) {
        B = Sf;
      }
      if (rejump) {
        A = q.AJ;
        T = q.TJ;
      }
    }
    if (act) q.S = add128(mul128(B, A), T);
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

template <int NR, int MT>
__device__ __forceinline__ void frag_matvec(const double (&F)[MT][NR], const double (&x)[NR], mfma_acc (&acc)[MT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < NR; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = DH_MFMA_F64(F[mt][s], x[s], acc[mt]);
}

// log-likelihood of the walker's v (quarter vector per sub-lane); every sub-lane returns the same bits
template <int NR, int MT, int KIND>
__device__ __forceinline__ double loglike_quad(const ProblemDev& P, int n, int t, const double (&v)[NR],
                                               const double* sprec, int lane, double* col) {
  cdptr lp = as_const(P.like_par);
  const int lid = like_of<KIND>(P);
  if (lid == LIKE_GAUSS_PREC) {
    // q = v . (P v): P v of the wave's 16 walkers as one matrix product, fragments of P from LDS
    mfma_acc w[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) w[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NR; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) w[mt] = DH_MFMA_F64(sprec[(mt * NR + s) * 64 + lane], v[s], w[mt]);
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
    for (int r = 0; r < NR; ++r) col[(4 * r + t) * 64] = v[r];
#pragma unroll 1
    for (int r = 0; r < NR; ++r)
      if (4 * r + t < n) prod *= cos((2.0 * tmax * col[(4 * r + t) * 64] - tmax) / 2.0);
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
  // (PRIOR_NORMAL is not built here: ocml's erfcinv next to the resident fragments spills ~160 VGPRs;
  // rwalk_launch_runs keeps such problems on the lane-per-walker kernel)
  const int pid = prior_of<KIND>(P);
  if (pid == PRIOR_AFFINE) {
    cdptr pp = as_const(P.prior_par);
    const double a = pp[0], b = pp[1];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (4 * r + t < n) ? a * (2.0 * u[r] - 1.0) + b : 0.0;
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = (4 * r + t < n) ? u[r] : 0.0;
  }
}

// generic_random_walk (internal_samplers.py:866-986) for ndim == ncdim, no periodic / reflective
// coordinates: four lanes per walker.  Workgroup = 4 wavefronts = 64 walkers.
template <int NR, int KIND, int RNG>
__global__ void __launch_bounds__(256) rwalkq_kernel(RwalkQArgs a) {
  constexpr int MT = (4 * NR + 15) / 16;
  __shared__ ZigQ zig;
  __shared__ double items[(4 * NR + 1) * 64];  // [item][walker slot]: a step's normals and its uniform
  __shared__ double sprec[MT * NR * 64];       // MFMA fragments of the precision matrix
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
      zig.fi[i] = __longlong_as_double((long long)a.zfi[i]);
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
    const int run = wi / a.wpr;
    on = a.run_mode[run] == a.my_mode;
    loglstar = a.run_loglstar[run];
    scale = a.run_scale[run];
  }
  // a wavefront leaves only as a whole: the matrix instructions take operands from all 64 lanes
  if (!__any(on)) return;

  double u[NR], up[NR], dr[NR], vv[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) u[r] = (4 * r + t < n) ? a.u0[(size_t)wi * n + 4 * r + t] : 0.5;
  QuadPcg q;
  hiprandStatePhilox4_32_10_t ph;
  if constexpr (RNG == RNGQ_PCG64) q.init(a.rng_in + (size_t)wi * 4, t);
  const int nb = (n + 3) >> 2;           // hiprand_normal4 blocks per step
  const int ph_stride = 4 * nb + 2;      // 32-bit draws per step of the lane-per-walker Philox kernel

  const int my_frame = a.axes_idx ? a.axes_idx[wi] : 0;
  const int f0 = __builtin_amdgcn_readfirstlane(my_frame);
  const bool uni = __all(my_frame == f0);
  double F[MT][NR];
  load_frags<NR, MT>(a.axes + (size_t)f0 * n * n, n, j, t, F);

  int nacc = 0, nrej = 0;
  double logl_cur = 0.0;
  const double inv_n = 1.0 / (double)n;

#pragma unroll 1
  for (int step = 0; step < a.walks; ++step) {
    // randsphere (bounding.py:1288-1297): n normals, one uniform
    if constexpr (RNG == RNGQ_PCG64) {
      quad_draw_step(q, &zig, items, slot, j, t, n);
    } else {
      // the walker's Philox subsequence exactly as walk.hip consumes it (per step: nb blocks of four
      // normals, one uniform double), block b drawn by sub-lane b & 3
      for (int b = t; b <= nb; b += 4) {
        hiprand_init(a.ph_seed, a.ph_seq0 + (unsigned long long)wi,
                     a.ph_offset + (unsigned long long)step * ph_stride + 4ull * b, &ph);
        if (b < nb) {
          const float4 zf = hiprand_normal4(&ph);
          items[(4 * b) * 64 + slot] = (double)zf.x;
          if (4 * b + 1 < n) items[(4 * b + 1) * 64 + slot] = (double)zf.y;
          if (4 * b + 2 < n) items[(4 * b + 2) * 64 + slot] = (double)zf.z;
          if (4 * b + 3 < n) items[(4 * b + 3) * 64 + slot] = (double)zf.w;
        } else {
          items[n * 64 + slot] = hiprand_uniform_double(&ph);
        }
      }
    }
    wave_sync();
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      dr[r] = (4 * r + t < n) ? items[(4 * r + t) * 64 + slot] : 0.0;
      ss = fma(dr[r], dr[r], ss);
    }
    const double ur = items[n * 64 + slot];
    wave_sync();
    ss = grp_sum(ss);
    // scale * ur^(1/n) / |dr| (bounding.py:1295-1296).  This is per-walker scalar work that all four
    // sub-lanes repeat, so it is kept short: exp(log(ur) / n) instead of ocml's double-double pow (|log
    // ur| / n is O(1): the product costs no accuracy), 1 / sqrt by v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(ss);
    y = y * fma(-0.5 * ss * y, y, 1.5);
    y = y * fma(-0.5 * ss * y, y, 1.5);
    const double fac = scale * (exp(inv_n * log(ur)) * y);
    // du = axes @ dr on the matrix cores; walkers of a wave on different frames: one product per frame
    mfma_acc acc[MT];
    if (uni) {
      frag_matvec<NR, MT>(F, dr, acc);
    } else {
      bool done = false;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
      for (;;) {
        const uint64_t rem = __ballot(!done);
        if (!rem) break;
        const int cur = __shfl(my_frame, __ffsll((long long)rem) - 1);
        load_frags<NR, MT>(a.axes + (size_t)cur * n * n, n, j, t, F);
        mfma_acc tmp[MT];
        frag_matvec<NR, MT>(F, dr, tmp);
        if (my_frame == cur) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = tmp[mt];
          done = true;
        }
      }
    }
    double lo = 0.5, hi = 0.5;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      up[r] = (4 * r + t < n) ? fma(fac, acc[r >> 2][r & 3], u[r]) : 0.5;
      lo = fmin(lo, up[r]);
      hi = fmax(hi, up[r]);
    }
    // unitcheck (utils.py:1036-1050) over the four quarters of the walker
    const bool inside_q = (lo > 0.0) && (hi < 1.0);
    const bool inside = grp_bits(__ballot(inside_q), j) == 0x0001000100010001ull;
    // a proposal outside the cube is counted as a call and a reject, no likelihood evaluated; here the
    // evaluation runs anyway (the matrix instruction serves the whole wave) and its verdict is ignored
    prior_quad<NR, KIND>(a.prob, n, t, up, vv);
    const double ll = loglike_quad<NR, MT, KIND>(a.prob, n, t, vv, sprec, lane, items + slot);
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
  const double ll0 = loglike_quad<NR, MT, KIND>(a.prob, n, t, vv, sprec, lane, items + slot);
  if (nacc == 0) logl_cur = ll0;
  if (live && on) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (4 * r + t < n) {
        a.u[(size_t)w * n + 4 * r + t] = u[r];
        a.v[(size_t)w * n + 4 * r + t] = vv[r];
      }
    if (t == 0) {
      a.logl[w] = logl_cur;
      a.nacc[w] = nacc;
      a.nrej[w] = nrej;
      if (RNG == RNGQ_PCG64 && a.rng_out) {
        const U128 b = q.base();
        uint64_t* o = a.rng_out + (size_t)w * 4;
        o[0] = b.hi;
        o[1] = b.lo;
        o[2] = q.inc.hi;
        o[3] = q.inc.lo;
      }
    }
  }
}

}  // namespace

namespace dh {

// Eligible launches (rwalk_launch_runs decides): ndim == ncdim in 9..32, no boundary conditions, fused
// likelihood, affine or identity prior.  Returns DH_OK after enqueueing on the context's stream.
int rwalkq_launch(dh_ctx* ctx, const ProblemDev& prob, int k, int ndim, const double* u0, const double* axes, int m,
                  const int32_t* axes_idx, double scale, double loglstar, int walks, const uint64_t* rng, double* u,
                  double* v, double* logl, int32_t* naccept, int32_t* nreject, uint64_t* rng_out,
                  const double* run_loglstar, const double* run_scale, const int* run_mode, int wpr, int my_mode,
                  const PhiloxKey* philox) {
  RwalkQArgs a;
  a.prob = prob;
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
  a.run_loglstar = run_loglstar;
  a.run_scale = run_scale;
  a.run_mode = run_mode;
  a.wpr = wpr;
  a.my_mode = my_mode;
  a.ph_seed = philox ? philox->seed : 0;
  a.ph_seq0 = philox ? philox->seq0 : 0;
  a.ph_offset = philox ? philox->offset : 0;
  const dim3 grid((k + 63) / 64), block(256);
  const int kind = problem_kind(prob.like_id, prob.prior_id) == KIND_PREC_AFFINE ? KIND_PREC_AFFINE : KIND_GENERIC;
  const int nr = ndim <= 16 ? 4 : ndim <= 28 ? 7 : 8;
#define L(NRR, KK)                                                                                     \
  do {                                                                                                 \
    if (philox)                                                                                        \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_PHILOX>), grid, block, 0, ctx->stream, a);       \
    else                                                                                               \
      hipLaunchKernelGGL((rwalkq_kernel<NRR, KK, RNGQ_PCG64>), grid, block, 0, ctx->stream, a);        \
  } while (0)
#define X(NRR)                         \
  if (nr == NRR) {                     \
    if (kind == KIND_PREC_AFFINE)      \
      L(NRR, KIND_PREC_AFFINE);        \
    else                               \
      L(NRR, KIND_GENERIC);            \
  }
  X(4) X(7) X(8)
#undef X
#undef L
  return hip_ok(ctx, hipGetLastError(), "rwalkq launch") ? DH_OK : DH_ERR_HIP;
}

}  // namespace dh
