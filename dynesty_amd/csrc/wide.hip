// Wide-D path (D up to 512): BASELINE config C4 is 200-D, where neither a
// walker (3 x 200 doubles) nor a D x D matrix (320 KB) fits a lane's registers
// or one CU's LDS.  Mapping changes accordingly:
//   * proposals: one WAVEFRONT per walker, lanes stride over the dimensions,
//     reductions are DPP wave reductions; the walker's PCG64 stream is drawn
//     lane-parallel (rng_pcg64.h: wave_normals / wave_doubles); the frame
//     products of the walkers of a workgroup are one GEMM on the matrix cores
//     (wg_frame_gemm); uniform sampling inside ellipsoids: wide_unif_kernel.
//   * Ellipsoid.update: partial sums / Gram matrices / Mahalanobis maxima over
//     up to 32 workgroups per run, the eigen-decomposition by one-sided block
//     Jacobi over several workgroups (wide_eig_kernel), the sequential rest on
//     one 1024-thread workgroup; a single-launch form with a two-sided Jacobi
//     on L2-resident matrices remains as the fallback (wide_single_kernel).
//   * MultiEllipsoid.update: the recursion of _bounding_ellipsoids on the host,
//     every node's work on the device (wide_multi_launch).
//   * membership: one workgroup per candidate point.
// Semantics and citations are those of the register-resident kernels
// (walk.hip / walk2.hip / rebuild.hip / bound.hip).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ctx.h"
#include "eig_wave.h"
#include "rng_gen.h"

using namespace dh;

namespace {

constexpr int kWideMaxD = 512;
constexpr int kRT = 1024;  // threads of the wide rebuild workgroup
constexpr int kTPMax = 64;  // points per LDS tile (32 / 16 where 64 rows of D | 1 doubles do not fit LDS)
typedef double wacc __attribute__((ext_vector_type(4)));
#define W_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Wave reductions through DPP (quad_perm / row mirrors inside each row of 16 lanes, row_bcast15/31
// across rows, readlane 63): ~20 VALU instructions instead of six ds_bpermute round trips.  The
// combination order is fixed, the result wave-uniform.
__device__ __forceinline__ double dpp_move(double v, const int ctrl_sel) {
  const long long b = __double_as_longlong(v);
  int lo = (int)b, hi = (int)(b >> 32);
  switch (ctrl_sel) {  // literal controls: the builtin needs immediates
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break;
    case 3: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xA, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xA, 0xF, false); break;
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x143, 0xC, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x143, 0xC, 0xF, false); break;
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_bcast63(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// rows 0/2 are masked out of the row_bcast steps (update_dpp returns the old value = v there): for
// sums the masked lanes must contribute 0 in those steps, so the partner value is taken only in the
// rows the mask enables
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move(v, 0);
  v += dpp_move(v, 1);
  v += dpp_move(v, 2);
  v += dpp_move(v, 3);  // every lane of a row: the row sum
  const int row = (threadIdx.x & 63) >> 4;
  const double b15 = dpp_move(v, 4);
  if (row & 1) v += b15;  // rows 1, 3: + previous row
  const double b31 = dpp_move(v, 5);
  if (row >= 2) v += b31;  // rows 2, 3: + (row 0 + row 1)
  return wave_bcast63(v);
}
__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, dpp_move(v, 0));
  v = fmin(v, dpp_move(v, 1));
  v = fmin(v, dpp_move(v, 2));
  v = fmin(v, dpp_move(v, 3));
  v = fmin(v, dpp_move(v, 4));
  v = fmin(v, dpp_move(v, 5));
  return wave_bcast63(v);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move(v, 0));
  v = fmax(v, dpp_move(v, 1));
  v = fmax(v, dpp_move(v, 2));
  v = fmax(v, dpp_move(v, 3));
  v = fmax(v, dpp_move(v, 4));
  v = fmax(v, dpp_move(v, 5));
  return wave_bcast63(v);
}

// prior_transform + loglikelihood for one walker spread over a wave:
// lane owns dims lane, lane+64, ...; u in LDS row `su` (D), v written to `sv`.
// For LIKE_GAUSS_PREC the precision matrix is read from global memory.
// tails: nullptr, or 256 doubles of LDS scratch of this wavefront -- then the Normal prior's ndtri routes the
// coordinates (round 4): the central approximation for all of them, the tail's (three times the work: logarithm,
// square root, second pair of polynomials) only for the ~15 % that need it, compacted over the lanes -- about one
// pass of the tail code for a 200-D walker instead of the four that evaluating both halves for every coordinate
// costs.  The same expressions either way: the same bits.
__device__ __forceinline__ double wide_logl(const ProblemDev& P, int D, const double* su, double* sv, int lane,
                                            double* tails = nullptr) {
  // prior
  if (P.prior_id == PRIOR_AFFINE) {
    const double a = P.prior_par[0], b = P.prior_par[1];
    for (int i = lane; i < D; i += 64) sv[i] = a * (2.0 * su[i] - 1.0) + b;
  } else if (P.prior_id == PRIOR_NORMAL && tails) {
    const double mu = P.prior_par[0], sg = P.prior_par[1];
    for (int base_i = 0; base_i < D; base_i += 256) {  // (wave-uniform trip count: the ballots need every lane)
      const int i0 = base_i + lane;
      double p[4];
      int slot[4];
      int ntail = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + 64 * q;
        const bool valid = i < D;
        p[q] = valid ? su[i] : 0.5;
        const bool tl = valid && !(fabs(p[q] - 0.5) <= 0.425);
        const unsigned long long m = __ballot(tl);
        slot[q] = tl ? ntail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) : -1;
        ntail += (int)__popcll(m);
        if (tl) tails[slot[q]] = p[q];
      }
      double o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ndtri_as241_central(p[q]);
      if (ntail > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < ntail; t0 += 64) {
          const int ti = t0 + lane;
          const double ptv = tails[ti < ntail ? ti : 0];
          const double pt = ti < ntail ? ptv : 0.25;
          bool far;
          double r = ndtri_as241_tail(pt, &far);
          if (far) r = ndtri_far(pt);
          if (ti < ntail) tails[ti] = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      {  // (round 6: the four reads unconditionally and together -- under their conditions each was a branch with an
         // LDS round trip and a full wait of its own, in every F evaluation)
        double tv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = tails[slot[q] >= 0 ? slot[q] : 0];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = slot[q] >= 0 ? tv[q] : o[q];
      }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + 64 * q < D) sv[i0 + 64 * q] = mu + sg * o[q];
    }
  } else if (P.prior_id == PRIOR_NORMAL) {
    const double mu = P.prior_par[0], sg = P.prior_par[1];
    for (int i0 = lane; i0 < D; i0 += 256) {  // 4 coordinates per lane at a time: independent chains
      double p[4], o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) p[q] = (i0 + 64 * q < D) ? su[i0 + 64 * q] : 0.5;
      ndtri_n<4>(p, o);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i0 + 64 * q < D) sv[i0 + 64 * q] = mu + sg * o[q];
    }
  } else {
    for (int i = lane; i < D; i += 64) sv[i] = su[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (P.like_id == LIKE_GAUSS_PREC) {
    const double* A = P.like_par + 1;
    double q = 0.0;
    for (int i = lane; i < D; i += 64) {
      double r = 0.0;
      const double* row = A + (size_t)i * D;
      for (int j = 0; j < D; ++j) r = fma(row[j], sv[j], r);
      q = fma(sv[i], r, q);
    }
    return P.like_par[0] - 0.5 * wave_sum(q);
  } else if (P.like_id == LIKE_EGGBOX) {
    const double tmax = P.like_par[0];
    // product over dims in index order (lane 0 only: D is small for eggbox)
    double prod = 1.0;
    for (int i = 0; i < D; ++i) prod *= cos((2.0 * tmax * sv[i] - tmax) / 2.0);
    const double b = 2.0 + prod, b2 = b * b;
    return b2 * b2 * b;
  } else {
    double q = 0.0;
    for (int i = lane; i < D; i += 64) q = fma(sv[i], sv[i], q);
    return P.like_par[0] - 0.5 * wave_sum(q);
  }
}

// F behind a call (round 4).  Inlined into wide_walk_kernel the evaluation -- four AS 241 `ndtri` chains per lane for
// the Normal prior, every fused likelihood -- shares one register allocation with the slice state machine, the
// generator and the frame product's sixteen requests in flight: 922 VGPRs spilled, ~900 scratch accesses inside the
// evaluation itself, i.e. in the hot loop (1 492 B of scratch per lane).  Behind a non-inlined call it
// gets an allocation of its own (the sort_slots lesson of ns.hip) and the caller keeps its state in callee-saved
// registers.  The arguments are scalars, not the kernel's argument block: a struct whose address escapes into a
// real call lives on every lane's stack.
__device__ __attribute__((noinline)) double wide_logl_call(int like_id, int prior_id, const double* like_par,
                                                           const double* prior_par, int D, const double* su,
                                                           double* sv, int lane, double* tails = nullptr) {
  ProblemDev P;
  P.like_id = like_id;
  P.prior_id = prior_id;
  P.ndim = D;
  P.like_par = like_par;
  P.prior_par = prior_par;
  P.prec_t = nullptr;
  return wide_logl(P, D, su, sv, lane, tails);
}
#ifdef DH_WIDE_F_CALL
#define WIDE_F(prob, D, su, sv, lane) \
  wide_logl_call((prob).like_id, (prob).prior_id, (prob).like_par, (prob).prior_par, (D), (su), (sv), (lane), wtails)
#else
#define WIDE_F(prob, D, su, sv, lane) wide_logl((prob), (D), (su), (sv), (lane), wtails)
#endif

// F(x) = logl(ptform(u + x * dir)) of the slice samplers from REGISTERS (round 4): for the iid Normal likelihood
// and D <= 256 a lane keeps its (at most four) coordinates of u and of the direction in registers for the whole
// slice, so an evaluation touches LDS only for ndtri's compacted tails -- no proposal vector written and read back,
// no v vector, no fences around them -- and needs one ballot and one wave reduction (the cube check, the sum of
// squares) instead of three reductions.  Same arithmetic per coordinate as wide_logl's: u' = fma(x, d, u),
// the same prior expressions, v^2 summed per lane over its coordinates in index order and then across the wave
// (wide_logl sums lane-wise in the same order), so a walker's path is the same to the last bit.
// Returns -inf outside the unit cube (generic_slice_step's unitcheck, internal_samplers.py:1075-1100).
__device__ __forceinline__ double wide_F_regs(const ProblemDev& P, int D, const double (&ur)[4], const double (&dr)[4],
                                              double x, int lane, double* tails) {
  double un[4];
  bool inside = true;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    un[q] = fma(x, dr[q], ur[q]);
    inside = inside && (!(lane + 64 * q < D) || (un[q] > 0.0 && un[q] < 1.0));
  }
  if (!__all(inside)) return -INFINITY;  // one ballot instead of the minimum and the maximum over the wave
  double v[4];
  if (P.prior_id == PRIOR_AFFINE) {
    const double a = P.prior_par[0], b = P.prior_par[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = a * (2.0 * un[q] - 1.0) + b;
  } else if (P.prior_id == PRIOR_NORMAL) {
    const double mu = P.prior_par[0], sg = P.prior_par[1];
    int slot[4], ntail = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool valid = lane + 64 * q < D;
      const double p = valid ? un[q] : 0.5;
      un[q] = p;
      const bool tl = valid && !(fabs(p - 0.5) <= 0.425);
      const unsigned long long m = __ballot(tl);
      slot[q] = tl ? ntail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) : -1;
      ntail += (int)__popcll(m);
      if (tl) tails[slot[q]] = p;
    }
    double o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = ndtri_as241_central(un[q]);
    if (ntail > 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int t0 = 0; t0 < ntail; t0 += 64) {
        const int ti = t0 + lane;
        const double ptv = tails[ti < ntail ? ti : 0];
        const double pt = ti < ntail ? ptv : 0.25;
        bool far;
        double r = ndtri_as241_tail(pt, &far);
        if (far) r = ndtri_far(pt);
        if (ti < ntail) tails[ti] = r;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      {
        double tv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = tails[slot[q] >= 0 ? slot[q] : 0];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = slot[q] >= 0 ? tv[q] : o[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = mu + sg * o[q];
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = un[q];
  }
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < D) s = fma(v[q], v[q], s);
  return P.like_par[0] - 0.5 * wave_sum(s);
}

struct WideWalkArgs {
  ProblemDev prob;
  int k, ndim, ncdim, m, iters;  // iters = walks or slices
  int kind;                      // 0 rwalk, 1 rslice, 2 slice
  int doubling0;
  double scale, loglstar;
  const double* u0;
  const double* axes_t;  // m x D x D, transposed (prep): AT[j*D + i] = axes[i][j]
  const int32_t* axes_idx;
  const int8_t* bc;
  const uint64_t* rng_in;
  double* u;
  double* v;
  double* logl;
  int32_t* c0;  // rwalk: naccept ; slice: ncalls
  int32_t* c1;  // rwalk: nreject ; slice: nexpand
  int32_t* c2;  // slice: ncontract
  int32_t* flags;
  uint64_t* rng_out;
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  int dbg;  // DH_WIDE_PROF=1: wave 0 prints its cycle split (normals / mat-vec / F evaluations)
  int lds_ws;  // doubles per wave region in LDS (4 D rounded up to odd)
  // ensemble form (ns.hip): walker w belongs to run w / wpr; per-run threshold / scale / doubling flag, and only
  // the runs whose mode is my_mode are served (all null / 0: the plain batch)
  const double* run_loglstar;
  const double* run_scale;
  const int* run_mode;
  const int* run_doubling;
  int wpr, my_mode;
  PhiloxKey ph;  // RNG_PHILOX
};

__device__ __forceinline__ void lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// The proposal frame applied to the vectors of ALL walkers of the workgroup at once:
// out_w[i] = scale * sum_k AT[k*n + i] * in_w[k]  =  one (n x n)(n x wpw) product on the matrix cores.
// One wavefront per walker alone re-read the whole frame (320 KB at D = 200) from L2 for every
// direction: 57 % of the C4 kernel.  Here the frame is read once per workgroup and direction
// round; the walkers of a workgroup meet at two barriers per round (they need their directions at
// the same point of the algorithm: once per rwalk step / rslice slice).
//   lds: per-wave regions of `ws` doubles; in/out = offsets of the vectors inside a region.
//   ncols walkers (columns) starting at region `lds`, worked on by `nw` wavefronts (this one = `wv`).
// A lone wavefront uses it on its own region (ncols = nw = 1) when the walkers of a workgroup sit on
// different frames: the arithmetic per column is the same, so a walker's path does not depend on
// the company it keeps.
// On gfx950 the fp64 matrix rate equals the vector rate (a 16x16x4 instruction takes 64+ cycles),
// and a workgroup has at most 4 walkers -- 4 of the 16 columns of that tile.  The 4x4x4 form
// (v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products, 17-24 cycles) fits exactly: the four
// blocks are four groups of 4 frame rows, all multiplied by the same 4 (k) x 4 (walker) block.
// Operand lanes (measured, tools/micro/mfma_f64_shapes.hip): A: block (l>>2)&3, i = l&3, k = l>>4
// -- i.e. frame row l&15, k = l>>4, as for the 16x16x4 form; B: k = l>>4, walker l&3 (any block);
// D: block (l>>2)&3, i = l>>4, walker l&3 -- one value per lane.
// A lane loads TWO neighbouring rows of the frame per request (16 B): rows 2 (l&15) and 2 (l&15) + 1
// of a 32-row span feed two tiles (even rows / odd rows of the span); 16 requests are in flight per
// lane and four accumulator chains (tile x k-step parity) keep the matrix pipe busy.
#define W_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
typedef double dv2 __attribute__((ext_vector_type(2)));
template <int NB>
__device__ __forceinline__ void gemm_batch(const __attribute__((address_space(1))) double*& ap, const double*& bp,
                                           size_t step, bool iv0, bool iv1, bool cv, double (&acc)[4]) {
  typedef const __attribute__((address_space(1))) dv2* g2ptr;
  dv2 fa[NB];
  double fb[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fa[q] = *(g2ptr)(ap + (size_t)q * step);
#pragma unroll
  for (int q = 0; q < NB; ++q) fb[q] = bp[4 * q];
  ap += NB * step;
  bp += 4 * NB;
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    // rows past n and walkers past ncols read clamped (finite) operands; their results are never stored
    acc[2 * (q & 1)] = W_MFMA4(fa[q].x, fb[q], acc[2 * (q & 1)]);
    acc[2 * (q & 1) + 1] = W_MFMA4(fa[q].y, fb[q], acc[2 * (q & 1) + 1]);
  }
}

__device__ __forceinline__ void wg_frame_gemm(const double* __restrict__ AT, int n, double* lds, int ws, int off_in,
                                          int off_out, double scale, int ncols, int wv, int nw) {
  typedef const __attribute__((address_space(1))) double* gptr;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4, lw = lane & 3;
  const int nsp = __builtin_amdgcn_readfirstlane((n + 31) >> 5), kfull = __builtin_amdgcn_readfirstlane(n >> 2);
  const bool cv = lw < ncols;
  const bool even = (n & 1) == 0;  // 16-byte requests need 16-byte aligned rows
  const double* inw = lds + (size_t)(cv ? lw : 0) * ws + off_in + lk;
  const size_t step = (size_t)4 * n;
  for (int sp = wv; sp < nsp; sp += nw) {
    const int i0 = sp * 32 + 2 * lr;
    const bool iv0 = i0 < n, iv1 = i0 + 1 < n;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const double* bp = inw;
    if (even) {
      gptr ap = (gptr)AT + (size_t)lk * n + (iv0 ? i0 : 0);
      int ks = 0;
#pragma unroll 1
      for (; ks + 16 <= kfull; ks += 16) gemm_batch<16>(ap, bp, step, iv0, iv1, cv, acc);
#pragma unroll 1
      for (; ks + 2 <= kfull; ks += 2) gemm_batch<2>(ap, bp, step, iv0, iv1, cv, acc);
      if (ks < kfull) gemm_batch<1>(ap, bp, step, iv0, iv1, cv, acc);
      if (n & 3) {
        const bool kv = kfull * 4 + lk < n;
        typedef const __attribute__((address_space(1))) dv2* g2ptr;
        dv2 fa = {0.0, 0.0};
        if (kv) fa = *(g2ptr)ap;
        const double b = (kv && cv) ? *bp : 0.0;
        acc[0] = W_MFMA4(iv0 ? fa.x : 0.0, b, acc[0]);
        acc[1] = W_MFMA4(iv1 ? fa.y : 0.0, b, acc[1]);
      }
    } else {
      gptr a0 = (gptr)AT + (size_t)lk * n + (iv0 ? i0 : 0);
      gptr a1 = (gptr)AT + (size_t)lk * n + (iv1 ? i0 + 1 : 0);
      for (int ks = 0; ks * 4 < n; ++ks) {
        const bool kv = ks * 4 + lk < n;
        const double f0 = (kv && iv0) ? *a0 : 0.0, f1 = (kv && iv1) ? *a1 : 0.0;
        const double b = (kv && cv) ? *bp : 0.0;
        a0 += step;
        a1 += step;
        bp += 4;
        acc[0] = W_MFMA4(f0, b, acc[0]);
        acc[1] = W_MFMA4(f1, b, acc[1]);
      }
    }
    if (cv) {
      // D: frame row 4 * ((l>>2)&3) + (l>>4) of the tile, walker l&3
      double* outw = lds + (size_t)lw * ws + off_out;
      const int row = sp * 32 + 2 * (4 * ((lane >> 2) & 3) + lk);
      if (row < n) outw[row] = (acc[0] + acc[2]) * scale;
      if (row + 1 < n) outw[row + 1] = (acc[1] + acc[3]) * scale;
    }
  }
}

enum SlicePhase { SL_LEFT0, SL_RIGHT0, SL_OUT_L, SL_OUT_R, SL_DBL, SL_SHRINK, SL_ACC, SL_DONE };
constexpr int kWalkMaxWaves = 4;  // walkers per workgroup: one wavefront per SIMD keeps the full register file per walker
// KIND: 0 rwalk, 1 rslice, 2 slice, 3 unit cube -- one instantiation each.  Two workgroups per CU
// (256 VGPRs): a lone wavefront issues a v_fma_f64 only every 8.5 cycles (tools/micro/
// mfma_f64_shapes.hip), so a second wavefront per SIMD nearly doubles the fp64 throughput of the F
// evaluations once there are more than 1024 walkers; the spills this costs are outside the hot loops.
// (round 6) The cycle split of DH_WIDE_PROF=1 is taken only when it is asked for: every F evaluation, direction round and
// frame product of every walker read the clock twice -- s_memtime and, with it, a wait for every outstanding LDS
// operation of the wavefront -- whether anybody looked at the numbers or not.
#define WCLK() (a.dbg ? clock64() : 0ll)
template <int KIND, int RNG>
__global__ void __launch_bounds__(64 * kWalkMaxWaves, 2) wide_walk_kernel(WideWalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ ZigLds zig;
  __shared__ int sframe[16];
  __shared__ double tails_all[kWalkMaxWaves][256];  // ndtri's tail coordinates of a wavefront, compacted (wide_logl)
  if constexpr (RNG == RNG_PCG64) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpw = blockDim.x >> 6;
  double* wtails = tails_all[wv];
  // XCD-aware workgroup order: workgroups are dealt to the 8 XCDs round-robin, each XCD with an L2 of its own, and
  // the walkers of a run all read the run's frame (320 KB at D = 200) for every direction.  With the plain order
  // every XCD reads every run's frame; here XCD x gets a contiguous eighth of the walkers, i.e. two of sixteen runs.
  const int nblk = gridDim.x;
  const int bid = (nblk & 7) == 0 ? (int)(blockIdx.x & 7) * (nblk >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int wq = bid * wpw + wv;
  // ghost = barriers only, no results: the padding wavefronts of the last workgroup, and (ensemble form) the
  // walkers of a run that is not in the mode this launch serves
  const int w = wq >= a.k ? a.k - 1 : wq;
  const int run = __builtin_amdgcn_readfirstlane(a.wpr > 0 ? w / a.wpr : 0);
  const bool ghost = wq >= a.k || (a.run_mode && a.run_mode[run] != a.my_mode);
  if (__syncthreads_and(ghost ? 1 : 0)) return;  // nobody in this workgroup has work
  const double loglstar = a.run_loglstar ? a.run_loglstar[run] : a.loglstar;
  const double scale = a.run_scale ? a.run_scale[run] : a.scale;
  const int D = a.ndim, nc = a.ncdim;
  const int ws = a.lds_ws;  // doubles per wave region (odd: conflict-free column reads in the GEMM)
  double* wbase = (double*)smem;
  double* su = wbase + (size_t)wv * ws;  // current point
  double* sp = su + D;                   // proposal / u_new
  double* sd = sp + D;                   // dr / direction
  double* sv = sd + D;                   // v
  int* sperm = (int*)(wbase + (size_t)wpw * ws) + (size_t)wv * ((D + 1) & ~1);
  if (a.u0)
    for (int i = lane; i < D; i += 64) su[i] = a.u0[(size_t)w * D + i];
  WaveGen<RNG> g;
  g.init(a.rng_in, (size_t)w, lane, &zig, a.ph);
  const int frame = __builtin_amdgcn_readfirstlane(a.axes_idx ? a.axes_idx[w] : 0);
  const double* AT = a.axes_t + (size_t)frame * nc * nc;  // unused when there are no frames (kind 3)
  // (walkers of different runs never share a frame index: frames are numbered run * m + ellipsoid)
  if (lane == 0) sframe[wv] = frame + (a.run_scale ? run * 0x100000 : 0);
  __syncthreads();
  bool coop = a.axes_t != nullptr;  // all walkers of the workgroup on one frame (and one run's scale): GEMM path
  for (int q = 0; q < wpw; ++q) coop = coop && sframe[q] == sframe[0];
  coop = __builtin_amdgcn_readfirstlane((int)coop) != 0;
  lds_sync();

  if (KIND == 3) {
    // ---- UnitCubeSampler.sample (internal_samplers.py:364-441) ----
    int ncall = 0;
    double ll = -INFINITY;
    for (;;) {
      g.doubles(su, D, lane);
      lds_sync();
      ll = WIDE_F(a.prob, D, su, sv, lane);
      ++ncall;
      lds_sync();
      if (ll > loglstar) break;
    }
    if (ghost) return;
    for (int i = lane; i < D; i += 64) {
      a.u[(size_t)w * D + i] = su[i];
      a.v[(size_t)w * D + i] = sv[i];
    }
    if (lane == 0) {
      a.logl[w] = ll;
      a.c0[w] = ncall;
      a.flags[w] = 0;
      g.store(a.rng_out, (size_t)w);
    }
    return;
  }
  if (KIND == 0) {
    // ---- rwalk (internal_samplers.py:866-1035) ----
    int nacc = 0, nrej = 0;
    double logl_cur = 0.0;
    for (int step = 0; step < a.iters; ++step) {
      if (nc < D) g.doubles(sp + nc, D - nc, lane);
      g.normals(sd, nc, lane);
      lds_sync();
      double ss = 0.0;
      for (int i = lane; i < nc; i += 64) ss = fma(sd[i], sd[i], ss);
      ss = wave_sum(ss);
      const double fac = scale * (pow(g.uniform(), 1.0 / (double)nc) / sqrt(ss));
      lds_sync();
      if (coop) {
        __syncthreads();
        wg_frame_gemm(AT, nc, wbase, ws, 2 * D, 3 * D, 1.0, wpw, wv, wpw);
        __syncthreads();
      } else {
        wg_frame_gemm(AT, nc, su, ws, 2 * D, 3 * D, 1.0, 1, 0, 1);
        lds_sync();
      }
      bool inside = true;
      for (int i = lane; i < nc; i += 64) sp[i] = fma(fac, sv[i], su[i]);
      lds_sync();
      for (int i = lane; i < D; i += 64) {
        const int b = a.bc ? a.bc[i] : 0;
        double x = sp[i];
        if (b == DH_BC_PERIODIC) x = x - floor(x);
        if (b == DH_BC_REFLECT) {
          const double m2 = x - 2.0 * floor(x * 0.5), m1 = x - floor(x);
          x = (m2 < 1.0) ? m1 : 1.0 - m1;
        }
        sp[i] = x;
        if (b == DH_BC_HARD)
          inside = inside && (x > 0.0) && (x < 1.0);
        else
          inside = inside && (x > -0.5) && (x < 1.5);
      }
      inside = __all(inside);
      lds_sync();
      if (!inside) {
        ++nrej;
        continue;
      }
      const double ll = WIDE_F(a.prob, D, sp, sv, lane);
      if (ll > loglstar) {
        for (int i = lane; i < D; i += 64) su[i] = sp[i];
        logl_cur = ll;
        ++nacc;
      } else {
        ++nrej;
      }
      lds_sync();
    }
    const double ll0 = WIDE_F(a.prob, D, su, sv, lane);
    if (nacc == 0) logl_cur = ll0;
    lds_sync();
    if (ghost) return;
    for (int i = lane; i < D; i += 64) {
      a.u[(size_t)w * D + i] = su[i];
      a.v[(size_t)w * D + i] = sv[i];
    }
    if (lane == 0) {
      a.logl[w] = logl_cur;
      a.c0[w] = nacc;
      a.c1[w] = nrej;
      g.store(a.rng_out, (size_t)w);
    }
    return;
  }

  // ---- rslice / slice (internal_samplers.py:593-855, 1038-1206) ----
  long long cy_n = 0, cy_m = 0, cy_f = 0, cy_g = 0, cy_t0 = WCLK();
  bool doubling = (a.run_doubling ? a.run_doubling[run] : a.doubling0) != 0, warn_set = false, failed = false;
  int ncall = 0, n_expand = 0, n_contract = 0;
  double logl_cur = 0.0;
  const double maxlen = sqrt((double)D) / 2.0;
  const int nsub = KIND == 1 ? 1 : D;
  for (int s = 0; s < a.iters && (!failed || KIND == 1); ++s) {
    if (KIND == 2) {
      // rstate.shuffle(arange(D)) -- sequential, every lane mirrors it
      for (int i = lane; i < D; i += 64) sperm[i] = i;
      lds_sync();
      for (int i = D - 1; i >= 1; --i) {
        const int j = (int)g.interval((uint64_t)i);
        if (lane == 0) {
          const int tmp = sperm[i];
          sperm[i] = sperm[j];
          sperm[j] = tmp;
        }
      }
      lds_sync();
    }
    for (int sub = 0; sub < nsub && (!failed || KIND == 1); ++sub) {
      if (KIND == 1) {
        // a failed walker keeps the workgroup's barriers company and does nothing else
        const long long c0_ = WCLK();
        if (!failed) {
          g.normals(sv, D, lane);  // sv as scratch for drhat
          lds_sync();
          double ss = 0.0;
          for (int i = lane; i < D; i += 64) ss = fma(sv[i], sv[i], ss);
          ss = wave_sum(ss);
          const double inv = 1.0 / sqrt(ss);
          lds_sync();
          for (int i = lane; i < D; i += 64) sv[i] = sv[i] * inv;
          lds_sync();
        }
        const long long c1_ = WCLK();
        cy_n += c1_ - c0_;
        if (coop) {
          __syncthreads();
          const long long cg_ = WCLK();
          wg_frame_gemm(AT, D, wbase, ws, 3 * D, 2 * D, scale, wpw, wv, wpw);
          cy_g += WCLK() - cg_;
          __syncthreads();
        } else if (!failed) {
          wg_frame_gemm(AT, D, su, ws, 3 * D, 2 * D, scale, 1, 0, 1);
        }
        cy_m += WCLK() - c1_;
        if (failed) continue;
      } else {
        const int idx = sperm[sub];
        const double* col = AT + (size_t)idx * D;
        for (int i = lane; i < D; i += 64) sd[i] = scale * col[i];
      }
      lds_sync();
      const double rand0 = g.uniform();
      double dl = 0.0;
      for (int i = lane; i < D; i += 64) dl = fma(sd[i], sd[i], dl);
      dl = sqrt(wave_sum(dl));
      const double dirnorm = dl > maxlen ? dl / maxlen : 1.0;
      for (int i = lane; i < D; i += 64) sd[i] = sd[i] / dirnorm;
      lds_sync();
      // generic_slice_step as a state machine around ONE evaluation site of F(x) (the phases of
      // walk2.hip's slice_kernel; control flow is wave-uniform here).  Eight inlined copies of F --
      // each with the prior and likelihood code -- made the kernel 570 KB of instructions: every
      // wavefront missed in the instruction cache all the time.
      // iid Normal likelihood, D <= 256: the walker's point and direction stay in registers for the slice
      const bool regF = a.prob.like_id == LIKE_GAUSS_IID && D <= 256 &&
                        (a.prob.prior_id == PRIOR_NORMAL || a.prob.prior_id == PRIOR_AFFINE || a.prob.prior_id == PRIOR_IDENTITY);
      double ureg[4] = {0.0, 0.0, 0.0, 0.0}, dreg[4] = {0.0, 0.0, 0.0, 0.0};
      if (regF) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (lane + 64 * q < D) {
            ureg[q] = su[lane + 64 * q];
            dreg[q] = sd[lane + 64 * q];
          }
      }
      double left = -rand0, right = 1.0 - rand0;
      double f_l = 0.0, f_r = 0.0;
      double Lw = 0.0, Rw = 0.0, fLw = 0.0, fRw = 0.0;
      double lhat = 0.0, rhat = 0.0, f_lhat = 0.0, f_rhat = 0.0, x1 = 0.0, logl_x1 = 0.0;
      bool Dflag = false, acc_right = false;
      int Kdbl = 1, nexp_step = 0;
      int phase = SL_LEFT0;
      double xq = left;
      while (phase != SL_DONE) {
        // F(xq): u_new = u + xq * direction; unitcheck; prior; likelihood
        const long long cf_ = WCLK();
        double f = -INFINITY;
        ++ncall;
        if (regF) {
          f = wide_F_regs(a.prob, D, ureg, dreg, xq, lane, wtails);
        } else {
          double lo = 2.0, hi = -1.0;
          for (int i = lane; i < D; i += 64) {
            const double un = fma(xq, sd[i], su[i]);
            sp[i] = un;
            lo = fmin(lo, un);
            hi = fmax(hi, un);
          }
          lo = wave_min(lo);
          hi = wave_max(hi);
          lds_sync();
          if (lo > 0.0 && hi < 1.0) {
            f = WIDE_F(a.prob, D, sp, sv, lane);
            lds_sync();
          }
        }
        cy_f += WCLK() - cf_;
        switch (phase) {
          case SL_LEFT0:
            f_l = f;
            phase = SL_RIGHT0;
            xq = right;
            break;
          case SL_RIGHT0:
            f_r = f;
            if (!doubling) {
              if (f_l > loglstar) {
                phase = SL_OUT_L;
                left -= 1.0;
                xq = left;
              } else if (f_r > loglstar) {
                phase = SL_OUT_R;
                right += 1.0;
                xq = right;
              } else {
                phase = SL_SHRINK;
              }
            } else {
              phase = SL_DBL;
            }
            break;
          case SL_OUT_L:
            f_l = f;
            ++nexp_step;
            if (f_l > loglstar) {
              left -= 1.0;
              xq = left;
            } else if (f_r > loglstar) {
              phase = SL_OUT_R;
              right += 1.0;
              xq = right;
            } else {
              phase = SL_SHRINK;
            }
            break;
          case SL_OUT_R:
            f_r = f;
            ++nexp_step;
            if (f_r > loglstar) {
              right += 1.0;
              xq = right;
            } else {
              phase = SL_SHRINK;
            }
            break;
          case SL_DBL:
            if (acc_right)
              f_r = f;
            else
              f_l = f;
            nexp_step += Kdbl;
            Kdbl *= 2;
            break;
          case SL_SHRINK: {
            ++n_contract;
            bool ok = f > loglstar;
            if (ok && doubling) {  // start Neal's acceptance test for x1 = xq
              x1 = xq;
              logl_x1 = f;
              lhat = Lw;
              rhat = Rw;
              f_lhat = fLw;
              f_rhat = fRw;
              Dflag = false;
              phase = SL_ACC;
              ok = false;
            }
            if (ok) {
              logl_cur = f;
              for (int i = lane; i < D; i += 64) su[i] = fma(xq, sd[i], su[i]);
              lds_sync();
              phase = SL_DONE;
            } else if (phase == SL_SHRINK) {
              if (xq < 0.0)
                left = xq;
              else if (xq > 0.0)
                right = xq;
              else {
                failed = true;
                phase = SL_DONE;
              }
            }
            break;
          }
          case SL_ACC:
            if (acc_right)
              f_rhat = f;
            else
              f_lhat = f;
            if (Dflag && loglstar >= f_lhat && loglstar >= f_rhat) {
              phase = SL_SHRINK;  // rejected: shrink towards the origin as for any failed proposal
              if (x1 < 0.0)
                left = x1;
              else if (x1 > 0.0)
                right = x1;
              else {
                failed = true;
                phase = SL_DONE;
              }
            }
            break;
          default:
            break;
        }
        // phases that choose their next abscissa after the switch
        if (phase == SL_DBL) {
          if (f_l > loglstar || f_r > loglstar) {
            if (g.uniform() < 0.5) {
              left -= (right - left);
              xq = left;
              acc_right = false;
            } else {
              right += (right - left);
              xq = right;
              acc_right = true;
            }
          } else {
            Lw = left;
            Rw = right;
            fLw = f_l;
            fRw = f_r;
            phase = SL_SHRINK;
          }
        }
        if (phase == SL_ACC) {
          if (rhat - lhat > 1.1) {
            const double M = (lhat + rhat) / 2.0;
            if ((0.0 < M && M <= x1) || (x1 < M && M <= 0.0)) Dflag = true;
            if (x1 < M) {
              rhat = M;
              xq = rhat;
              acc_right = true;
            } else {
              lhat = M;
              xq = lhat;
              acc_right = false;
            }
          } else {  // accepted
            logl_cur = logl_x1;
            for (int i = lane; i < D; i += 64) su[i] = fma(x1, sd[i], su[i]);
            lds_sync();
            phase = SL_DONE;
          }
        }
        if (phase == SL_SHRINK) xq = left + g.uniform() * (right - left);
      }
      n_expand += nexp_step;
      if (!doubling && nexp_step > 1000) {
        doubling = true;
        warn_set = true;
      }
    }
  }
  if (a.dbg && w == 0 && lane == 0)
    printf("wide_walk wave 0: total %lld | normals %lld | frame product %lld (GEMM itself %lld) | F %lld (ncall %d)\n",
           (long long)(WCLK() - cy_t0), cy_n, cy_m, cy_g, cy_f, ncall);
  (void)WIDE_F(a.prob, D, su, sv, lane);  // v of the returned point
  lds_sync();
  if (ghost) return;
  for (int i = lane; i < D; i += 64) {
    a.u[(size_t)w * D + i] = su[i];
    a.v[(size_t)w * D + i] = sv[i];
  }
  if (lane == 0) {
    a.logl[w] = logl_cur;
    a.c0[w] = ncall;
    a.c1[w] = n_expand;
    a.c2[w] = n_contract;
    a.flags[w] = (warn_set ? 1 : 0) | (failed ? 2 : 0);
    g.store(a.rng_out, (size_t)w);
  }
}

// ---- UniformBoundSampler.sample at wide D (internal_samplers.py:243-340) ----------------------
// One wavefront per walker: rand_choice over the ellipsoids, randsphere (lane-parallel normals),
// frame product, membership count q over all ellipsoids (strict, as MultiEllipsoid.within) with the
// 1/q acceptance, unitcheck, uniforms for the non-clustered coordinates, prior + likelihood -- the
// draw order of the narrow unif_kernel.  propose_only = the lock-step form (problem = -1).
struct WideUnifArgs {
  ProblemDev prob;
  int k, ndim, ncdim, m, propose_only;
  double loglstar;
  const double* ctrs;     // m x nc
  const double* axes_t;   // m x nc x nc transposed: AT[j*nc + i] = axes[i][j]
  const double* ams;      // m x nc x nc (symmetric)
  const double* cumprob;  // m
  const int8_t* bc;
  const uint64_t* rng_in;
  int64_t max_tries;
  double* u;
  double* v;
  double* logl;
  int32_t* ncalls;
  int32_t* flags;
  uint64_t* rng_out;
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  PhiloxKey ph;  // RNG_PHILOX
  // ensemble form (ns.hip, round 5): walker w belongs to run w / wpr; per-run threshold, only the runs whose mode is
  // my_mode are served; run r owns the ellipsoids [r * run_me, r * run_me + M_r) of ctrs / axes_t / ams / cumprob,
  // M_r = run_nells[r] (null: 1).  All null / 0: the plain batch.
  const double* run_loglstar;
  const int* run_mode;
  const int* run_nells;
  int wpr, my_mode, run_me;
};

#undef WCLK
template <int RNG>
__global__ void __launch_bounds__(64) wide_unif_kernel(WideUnifArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ ZigLds zig;
  if constexpr (RNG == RNG_PCG64) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int lane = threadIdx.x, w = blockIdx.x, D = a.ndim, nc = a.ncdim;
  double* sz = (double*)smem;  // nc: normals, then x - c
  double* sx = sz + nc;        // D : candidate
  double* sv = sx + D;         // D : v
  int M = a.m;
  size_t eb = 0;
  double loglstar = a.loglstar;
  if (a.run_mode) {
    const int run = w / a.wpr;
    if (a.run_mode[run] != a.my_mode) return;
    loglstar = a.run_loglstar[run];
    if (a.run_me > 0) {
      M = a.run_nells ? a.run_nells[run] : 1;
      eb = (size_t)run * a.run_me;
    }
  }
  const double* cumprob = a.cumprob ? a.cumprob + eb : nullptr;
  WaveGen<RNG> g;
  g.init(a.rng_in, (size_t)w, lane, &zig, a.ph);
  int ncall = 0, flags = 0;
  double logl_cur = 0.0;
  int64_t tries = 0;
  for (;;) {
    if (tries >= a.max_tries) {
      flags |= 2;
      break;
    }
    ++tries;
    if (M == 0) {  // unit cube: rstate.uniform(size=ndim)
      g.doubles(sx, D, lane);
      lds_sync();
      if (a.propose_only) break;
      const double ll0 = wide_logl(a.prob, D, sx, sv, lane);
      lds_sync();
      ++ncall;
      if (ll0 > loglstar) {
        logl_cur = ll0;
        break;
      }
      continue;
    }
    int idx = 0;
    if (M > 1) {  // rand_choice (bounding.py:1300-1308)
      const double xr = g.uniform();
      while (idx < M - 1 && cumprob[idx] < xr) ++idx;
    }
    g.normals(sz, nc, lane);
    lds_sync();
    double ss = 0.0;
    for (int i = 0; i < nc; ++i) ss = fma(sz[i], sz[i], ss);  // index order, as the narrow kernel
    const double fac = pow(g.uniform(), 1.0 / (double)nc) / sqrt(ss);
    const double* AT = a.axes_t + (eb + (size_t)idx) * nc * nc;
    const double* c = a.ctrs + (eb + (size_t)idx) * nc;
    for (int i = lane; i < nc; i += 64) {
      double r = 0.0;
      for (int j = 0; j < nc; ++j) r = fma(AT[(size_t)j * nc + i], sz[j], r);
      sx[i] = fma(fac, r, c[i]);
    }
    lds_sync();
    bool accept = true;
    if (M > 1) {
      int q = 0, qloose = 0;
      for (int e = 0; e < M; ++e) {
        const double* ce = a.ctrs + (eb + (size_t)e) * nc;
        const double* A = a.ams + (eb + (size_t)e) * nc * nc;
        for (int i = lane; i < nc; i += 64) sz[i] = sx[i] - ce[i];
        lds_sync();
        double part = 0.0;
        for (int i = lane; i < nc; i += 64) {
          double r = 0.0;
          for (int j = 0; j < nc; ++j) r = fma(A[(size_t)j * nc + i], sz[j], r);
          part = fma(sz[i], r, part);
        }
        const double quad = wave_sum(part);
        lds_sync();
        q += quad < 1.0 ? 1 : 0;
        qloose += quad <= 1.0 + 1e-3 ? 1 : 0;
      }
      if (q == 0) {
        q = qloose;
        if (q == 0) {
          flags |= 1;
          break;
        }
      }
      if (q > 1) accept = g.uniform() < (1.0 / (double)q);
    }
    if (!accept) continue;
    bool inside = true;
    for (int i = lane; i < nc; i += 64) {
      const int b = a.bc ? a.bc[i] : 0;
      const double x = sx[i];
      if (b == DH_BC_HARD)
        inside = inside && (x > 0.0) && (x < 1.0);
      else
        inside = inside && (x > -0.5) && (x < 1.5);
    }
    inside = __all(inside);
    if (!inside) continue;
    if (nc < D) {
      g.doubles(sx + nc, D - nc, lane);
      lds_sync();
    }
    if (a.propose_only) break;
    const double ll = wide_logl(a.prob, D, sx, sv, lane);
    lds_sync();
    ++ncall;
    if (ll > loglstar) {
      logl_cur = ll;
      break;
    }
  }
  for (int i = lane; i < D; i += 64) {
    a.u[(size_t)w * D + i] = sx[i];
    if (!a.propose_only) a.v[(size_t)w * D + i] = sv[i];
  }
  if (lane == 0) {
    if (!a.propose_only) {
      a.logl[w] = logl_cur;
      a.ncalls[w] = ncall;
    }
    a.flags[w] = flags;
    g.store(a.rng_out, (size_t)w);
  }
}

__global__ void __launch_bounds__(64)
    wide_eval_kernel(ProblemDev prob, int k, const double* __restrict__ u, double* v, double* logl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = prob.ndim, lane = threadIdx.x, w = blockIdx.x;
  double* su = (double*)smem;
  double* sv = su + D;
  for (int i = lane; i < D; i += 64) su[i] = u[(size_t)w * D + i];
  lds_sync();
  const double ll = wide_logl(prob, D, su, sv, lane);
  lds_sync();
  for (int i = lane; i < D; i += 64) v[(size_t)w * D + i] = sv[i];
  if (lane == 0) logl[w] = ll;
}

__global__ void wide_transpose_kernel(const double* __restrict__ in, int m, int d, double* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)m * d * d;
  if (t >= tot) return;
  const size_t f = t / ((size_t)d * d), r = t % ((size_t)d * d);
  const int j = (int)(r / d), i = (int)(r % d);
  out[t] = in[f * d * d + (size_t)i * d + j];
}

// ---------------------------------------------------------------------------
// membership, one workgroup (256 threads) per candidate point
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    wide_contains_kernel(const double* __restrict__ x, int k, int d, const double* __restrict__ ctrs,
                         const double* __restrict__ ams, int m, int mode, int32_t* count, uint64_t* mask,
                         double* quad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* dl = (double*)smem;  // d
  double* red = dl + d;        // 256
  const int p = blockIdx.x, t = threadIdx.x;
  int cnt = 0;
  const int nwords = (k + 63) / 64;
  for (int e = 0; e < m; ++e) {
    for (int i = t; i < d; i += 256) dl[i] = x[(size_t)p * d + i] - ctrs[(size_t)e * d + i];
    __syncthreads();
    const double* A = ams + (size_t)e * d * d;
    double q = 0.0;
    for (int i = t; i < d; i += 256) {
      double r = 0.0;
      const double* row = A + (size_t)i * d;
      for (int j = 0; j < d; ++j) r = fma(row[j], dl[j], r);
      q = fma(dl[i], r, q);
    }
    red[t] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) red[t] += red[t + s];
      __syncthreads();
    }
    q = red[0];
    __syncthreads();
    const bool in = mode == 0 ? (q < 1.0) : (sqrt(q) <= 1.0);
    cnt += in ? 1 : 0;
    if (t == 0) {
      if (quad) quad[(size_t)p * m + e] = q;
      if (mask && in) atomicOr((unsigned long long*)&mask[(size_t)e * nwords + p / 64], 1ull << (p & 63));
    }
  }
  if (t == 0) count[p] = cnt;
}

// ---------------------------------------------------------------------------
// Ellipsoid.update (bounding_ellipsoid, bounding.py:1387-1461) for wide D:
// one 1024-thread workgroup per run.
// ---------------------------------------------------------------------------
struct WideRebuildArgs {
  const double* pts;
  int n, d, runs;
  double prefactor;
  double* wsA;   // runs x d x d   Jacobi work
  double* wsV;   // runs x d x d
  double* wscov; // runs x d x d   working covariance
  int dbg;       // DH_WIDE_PROF=1: thread 0 of run 0 prints the cycle count of every phase
  double* wsW;   // runs x 4 x P x P  double-buffered Jacobi work (P = d rounded up to even)
  int phase;     // 0: whole rebuild in one launch; 2: everything after the covariance (centre in ctrs,
                 // covariance in wscov), taking its eigen-decomposition from wide_eig_kernel when eig_ok says so
  const double* lam_pre;  // runs x d   eigenvalues from wide_eig_kernel (vectors are in wsV)
  const int* eig_ok;      // runs
  // phase 2 with split_tail: stop before the Mahalanobis maximum (wide_fmax_part_kernel /
  // wide_finish_kernel take over) unless the covariance had to be regularised
  int tp;                    // points per LDS tile
  int split_tail, P, chunk;  // P chunks of `chunk` points per run
  double* meanpart;          // runs x P x d
  double* covpart;           // runs x P x d x d
  double* fmaxpart;          // runs x P
  int* fast;                 // runs: 1 = phase 2 left the tail to the split kernels
  double* lam_ws;            // runs x d
  int* order_ws;             // runs x d
  int* status;
  double* ctrs;
  double* covs;
  double* ams;
  double* axes;
  double* axlens;
  double* logvols;
  const int* active;  // runs or null: runs with active[run] == 0 are left untouched (ensemble loop)
};

__device__ double block_max_1024(double v, double* red) {
  const int t = threadIdx.x;
  v = wave_max(v);
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < kRT / 64; ++i) r = fmax(r, red[i]);
  __syncthreads();
  return r;
}

// Parallel-order cyclic Jacobi on global-memory (L2-resident) matrices, the formulation of
// rebuild.hip's jacobi_block for D up to 512: position-based tournament (pair k always sits at
// positions 2k, 2k+1; a round WRITES the rotated 2x2 blocks to where the circle method moves
// them, into the second buffer), so every index is loop-invariant, the block loads are
// coalesced pairs, and a round costs two workgroup barriers -- one after the m rotations are
// derived (by m threads, shared through LDS), one after the blocks are written.  The previous
// three-phase form (columns strided by D, integer divisions in the inner loops) cost ~39 us
// per round at D = 200; 10 sweeps x 199 rounds made the eigensolver 95 % of the 78 ms rebuild.
// Work buffers: W = 4 matrices of P x P (P = D rounded up to even): A0 | A1 | V0 | V1.
// On return the diagonal of A holds the eigenvalues (unsorted), the columns of V the vectors.
__device__ bool jacobi_global(double* A, double* V, double* W, int D, double* rc, double* rs, double* red) {
  const int t = threadIdx.x;
  const int m = (D + 1) / 2, P = 2 * m;
  const size_t PP = (size_t)P * P;
  double* wa[2] = {W, W + PP};
  double* wv[2] = {W + 2 * PP, W + 3 * PP};
  bool bad = false;
  double amax = 0.0;
  for (int e = t; e < D * D; e += kRT) {
    const double v = A[e];
    if (!isfinite(v)) bad = true;
    amax = fmax(amax, fabs(v));
  }
  if (__syncthreads_or(bad ? 1 : 0)) return false;
  if (D == 1) {
    if (t == 0) V[0] = 1.0;
    __syncthreads();
    return true;
  }
  amax = block_max_1024(amax, red);
  const int ex = amax > 0.0 ? ilogb(amax) + 1 : 0;  // power-of-two pre-scaling (see jacobi_rotation)
  for (int e = t; e < P * P; e += kRT) {
    const int i = e / P, j = e - i * P;
    wa[0][e] = (i < D && j < D) ? ldexp(A[(size_t)i * D + j], -ex) : 0.0;
    wv[0][e] = (i == j && i < D) ? 1.0 : 0.0;
  }
  __threadfence_block();
  __syncthreads();
  // block ownership: blocks blk = t + s * kRT, s < NS (m^2 <= 65 536 -> NS <= 64); the (row pair,
  // column pair) of a slot is recomputed each round from two running counters instead of a division
  const int nblk = m * m;
  int cur = 0;
  int dpos = (D & 1) ? D : -1;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double* Ac = wa[cur];
    double off = 0.0, dia = 0.0;
    for (int e = t; e < P * P; e += kRT) {
      const int i = e / P, j = e - i * P;
      const double v = Ac[e];
      if (i == j)
        dia = fma(v, v, dia);
      else
        off = fma(v, v, off);
    }
    off = wave_sum(off);
    dia = wave_sum(dia);
    if ((t & 63) == 0) {
      red[t >> 6] = off;
      red[16 + (t >> 6)] = dia;
    }
    __syncthreads();
    off = dia = 0.0;
    for (int i = 0; i < kRT / 64; ++i) {
      off += red[i];
      dia += red[16 + i];
    }
    __syncthreads();
    if (!(off > 1e-31 * dia)) break;
    for (int r = 0; r < P - 1; ++r) {
      const double* As = wa[cur];
      const double* Vs = wv[cur];
      double* Ad = wa[cur ^ 1];
      double* Vd = wv[cur ^ 1];
      // the m rotations of this round
      for (int k = t; k < m; k += kRT) {
        const size_t p0 = (size_t)(2 * k) * P + 2 * k;
        double c, sn;
        dh_eig::jacobi_rotation(As[p0], As[p0 + P + 1], As[p0 + 1], c, sn);
        rc[k] = c;
        rs[k] = sn;
      }
      __syncthreads();
      // all 2x2 blocks: rows J_r^T ., columns . J_c, written to the positions of the next round.
      // Four blocks per thread are loaded before any is stored (source and destination buffers
      // are distinct, but the compiler cannot know: without the batching every block waited
      // a full L2 round trip behind the stores of the previous one -- 14 us per round).
      constexpr int kBB = 4;
      for (int blk0 = t; blk0 < nblk; blk0 += kBB * kRT) {
        double2 xa[kBB], xb[kBB], va[kBB], vb[kBB];
        int kr[kBB], kc[kBB];
#pragma unroll
        for (int u = 0; u < kBB; ++u) {
          const int blk = blk0 + u * kRT;
          const int b2 = blk < nblk ? blk : 0;
          kr[u] = b2 / m;
          kc[u] = b2 - kr[u] * m;
          const size_t o0 = (size_t)(2 * kr[u]) * P + 2 * kc[u];
          xa[u] = *(const double2*)(As + o0);
          xb[u] = *(const double2*)(As + o0 + P);
          va[u] = *(const double2*)(Vs + o0);
          vb[u] = *(const double2*)(Vs + o0 + P);
        }
#pragma unroll
        for (int u = 0; u < kBB; ++u) {
          if (blk0 + u * kRT >= nblk) continue;
          const int r0 = 2 * kr[u], c0 = 2 * kc[u];
          const double cr = rc[kr[u]], sr = rs[kr[u]], cc = rc[kc[u]], sc = rs[kc[u]];
          const double y00 = cr * xa[u].x - sr * xb[u].x, y01 = cr * xa[u].y - sr * xb[u].y;
          const double y10 = sr * xa[u].x + cr * xb[u].x, y11 = sr * xa[u].y + cr * xb[u].y;
          double z00 = cc * y00 - sc * y01, z01 = sc * y00 + cc * y01;
          double z10 = cc * y10 - sc * y11, z11 = sc * y10 + cc * y11;
          if (r0 == c0) z01 = z10 = 0.0;  // the annihilated pair is exactly zero
          const int dra = dh_eig::jacobi_dest(r0, m), drb = dh_eig::jacobi_dest(r0 + 1, m);
          const int dca = dh_eig::jacobi_dest(c0, m), dcb = dh_eig::jacobi_dest(c0 + 1, m);
          Ad[(size_t)dra * P + dca] = z00;
          Ad[(size_t)dra * P + dcb] = z01;
          Ad[(size_t)drb * P + dca] = z10;
          Ad[(size_t)drb * P + dcb] = z11;
          Vd[(size_t)r0 * P + dca] = cc * va[u].x - sc * va[u].y;
          Vd[(size_t)r0 * P + dcb] = sc * va[u].x + cc * va[u].y;
          Vd[(size_t)(r0 + 1) * P + dca] = cc * vb[u].x - sc * vb[u].y;
          Vd[(size_t)(r0 + 1) * P + dcb] = sc * vb[u].x + cc * vb[u].y;
        }
      }
      if (dpos >= 0) dpos = dh_eig::jacobi_dest(dpos, m);
      cur ^= 1;
      __threadfence_block();
      __syncthreads();
    }
  }
  // copy out: position a (skipping the padding) -> column a' of V, diagonal of A
  const double* Af = wa[cur];
  const double* Vf = wv[cur];
  for (int e = t; e < P * P; e += kRT) {
    const int i = e / P, a2 = e - i * P;
    if (a2 == dpos || i >= D) continue;
    const int col = a2 - ((dpos >= 0 && a2 > dpos) ? 1 : 0);
    V[(size_t)i * D + col] = Vf[e];
    if (i == 0) A[(size_t)col * D + col] = ldexp(Af[(size_t)a2 * P + a2], ex);
  }
  __threadfence_block();
  __syncthreads();
  return true;
}

// stage points [base, base+cnt) of the run, centred on `mean`, into the LDS tile (row stride D|1)
__device__ __forceinline__ void wide_stage(const double* pts, int D, int base, int cnt, const double* mean,
                                           double* tile) {
  const int LD = D | 1;
  for (int e = threadIdx.x; e < cnt * D; e += kRT) {
    const int p = e / D, j = e - p * D;
    tile[p * LD + j] = pts[(size_t)(base + p) * D + j] - mean[j];
  }
  __syncthreads();
}

// out = inv * Xc^T Xc over the points [pbeg, pend) (one 1024-thread workgroup); with `mirror` both
// triangles are written, without only the 16x16 blocks (ib <= jb) -- the form the partial sums of the
// multi-workgroup path use.
__device__ void wide_cov_range(const double* pts, int D, int tp, int pbeg, int pend, const double* mean,
                               double* tile, double* out, double inv, bool mirror) {
  const int t = threadIdx.x, LD = D | 1;
    // ---- covariance (np.cov ddof=1) on the matrix cores: C = Xc^T Xc, upper 16x16 blocks ----
    // Block pairs (ib <= jb) are dealt round-robin to the 16 waves, kCovPairs per wave and pass
    // (accumulators stay in registers across all tiles of the pass); K = points, 16 MFMA steps
    // of 4 per 64-point tile.  The VALU form read two LDS operands per FMA (3.9 ms at 4000x200).
    {
      const int lane = t & 63, wv = t >> 6, lj = lane & 15, lk = lane >> 4;
      const int nbk = (D + 15) >> 4;
      const int npairs = nbk * (nbk + 1) / 2;
      constexpr int kCovPairs = 6;
      for (int pass0 = 0; pass0 < npairs; pass0 += kCovPairs * (kRT / 64)) {
        wacc acc[kCovPairs];
        int pib[kCovPairs], pjb[kCovPairs];
#pragma unroll
        for (int r = 0; r < kCovPairs; ++r) {
          acc[r] = (wacc){0.0, 0.0, 0.0, 0.0};
          const int pr = pass0 + r * (kRT / 64) + wv;
          // unrank pr -> (ib <= jb), row-major over the upper triangle of an nbk x nbk grid
          int ib = 0, rem = pr;
          while (ib < nbk && rem >= nbk - ib) {
            rem -= nbk - ib;
            ++ib;
          }
          pib[r] = pr < npairs ? ib : -1;
          pjb[r] = ib + rem;
        }
        for (int base = pbeg; base < pend; base += tp) {
          const int cnt = min(tp, pend - base);
          wide_stage(pts, D, base, cnt, mean, tile);
#pragma unroll
          for (int r = 0; r < kCovPairs; ++r) {
            if (pib[r] < 0) continue;
            const int ca = pib[r] * 16 + lj, cb = pjb[r] * 16 + lj;
            for (int p0 = 0; p0 < cnt; p0 += 4) {
              const int pp = p0 + lk;
              const bool pv = pp < cnt;
              const double fa = (pv && ca < D) ? tile[pp * LD + ca] : 0.0;
              const double fb = (pv && cb < D) ? tile[pp * LD + cb] : 0.0;
              acc[r] = W_MFMA(fa, fb, acc[r]);
            }
          }
          __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < kCovPairs; ++r) {
          if (pib[r] < 0) continue;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int i = pib[r] * 16 + lk + 4 * q4, j = pjb[r] * 16 + lj;
            if (i < D && j < D) {
              const double c = acc[r][q4] * inv;
              out[(size_t)i * D + j] = c;
              if (mirror) out[(size_t)j * D + i] = c;
            }
          }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
}

// max over the points [pbeg, pend) of delta^T am delta (per-thread partial maxima; the caller reduces)
__device__ double wide_fmax_range(const double* pts, int D, int tp, int pbeg, int pend, const double* mean,
                                  double* tile, double* part, const double* o_am) {
  const int t = threadIdx.x, LD = D | 1;
      // ---- fmax = max_p delta^T am delta on the matrix cores: Z = Xc AM per 64-point tile
      // (M = 4 point blocks, N = column blocks dealt to the waves, K = D in steps of 4), then the
      // row-wise dot Z . x, a 16-lane reduction and a fixed-order sum over the column blocks.
      // The VALU form issued one global load per FMA (17.5 ms at 4000 x 200).
      double best = -INFINITY;
      {
        const int lane = t & 63, wv = t >> 6, lj = lane & 15, lk = lane >> 4;
        const int nbk = (D + 15) >> 4, ksteps = (D + 3) >> 2;
        for (int base = pbeg; base < pend; base += tp) {
          const int cnt = min(tp, pend - base);
          wide_stage(pts, D, base, cnt, mean, tile);
          for (int jb = wv; jb < nbk; jb += kRT / 64) {
            const int jc = jb * 16 + lj;
            wacc z[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) z[mb] = (wacc){0.0, 0.0, 0.0, 0.0};
            for (int ks = 0; ks < ksteps; ++ks) {
              const int k = ks * 4 + lk;
              const bool kv = k < D;
              const double fb = (kv && jc < D) ? o_am[(size_t)k * D + jc] : 0.0;
#pragma unroll
              for (int mb = 0; mb < 4; ++mb) {
                const int pp = mb * 16 + lj;
                const double fa = (kv && pp < cnt) ? tile[pp * LD + k] : 0.0;
                z[mb] = W_MFMA(fa, fb, z[mb]);
              }
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const int pp = mb * 16 + lk + 4 * q4;
                double sacc = (pp < cnt && jc < D) ? z[mb][q4] * tile[pp * LD + jc] : 0.0;
                sacc += __shfl_xor(sacc, 1);
                sacc += __shfl_xor(sacc, 2);
                sacc += __shfl_xor(sacc, 4);
                sacc += __shfl_xor(sacc, 8);
                if (lj == 0) part[jb * 64 + pp] = sacc;
              }
            }
          }
          __syncthreads();
          if (t < cnt) {
            double sacc = 0.0;
            for (int jb = 0; jb < nbk; ++jb) sacc += part[jb * 64 + t];
            best = fmax(best, sacc);
          }
          __syncthreads();
        }
      }
  return best;
}

// ---------------------------------------------------------------------------
// Eigen-decomposition of the covariance by ONE-SIDED block Jacobi over several workgroups.
//
// jacobi_global above is two-sided: every round rewrites A and V (1.3 MB at D = 200) through ONE
// compute unit's memory pipeline -- 30 ms of the 37 ms rebuild at 4000 x 200.  Hestenes' form
// rotates COLUMNS only: with G = A (symmetric), right rotations that make the columns of G J
// mutually orthogonal give J = V (A V = V L has orthogonal columns), and a rotation needs just the
// two columns it touches: three dot products and two axpys.  So the columns are dealt in 2B blocks
// of b (<= 16) to B workgroups, each keeps its two blocks [G column | V column] in LDS and
// orthogonalises their pairs there (one wavefront per pair, b pairs at a time); between block
// rounds the blocks move to their next partner through global memory (agent-scope stores / loads,
// the barrier of rebuild.hip's parts -- workgroups may sit on different XCDs).  Block pairing is
// the circle method; pairs inside a block are met in the first round of every sweep.  A sweep
// without a rotation ends the iteration.  lam_k = v_k . (A v_k) (Rayleigh quotient of the final
// column), V is orthogonal by construction (it only ever sees rotations).
// LDS: 2b columns of 2D doubles.  Grid = runs x B, all resident (launch checks B * runs <= CUs).
struct WideEigArgs {
  const double* cov;  // runs x D x D
  double* lam;        // runs x D
  double* V;          // runs x D x D : V[i*D + k] = component i of vector k
  double* xbuf;       // runs x 2 x M x b x 2D   block exchange (two parities)
  int* bar;           // runs  (zeroed before launch)
  int* rot;           // runs x kEigMaxSweeps rotation counters (zeroed)
  int* ok;            // runs
  int D, B, b;
  int dbg;
  const int* active;  // see WideRebuildArgs
  // warm start (wide_warm_kernel): the previous rebuild's eigenvectors as rows of V0t and Wt = (cov V0)^T;
  // cold[run] != 0 (or null pointers): start from the identity
  const double* V0t;
  const double* Wt;
  const int* cold;
};
constexpr int kEigMaxSweeps = 30;

// Warm start of the one-sided Jacobi solver.  Between two consecutive rebuilds of a run the covariance of the live
// points barely moves (one queue fill replaces a few per cent of them), so the previous eigenvectors V0 nearly
// diagonalise the new matrix: Hestenes' iteration started from G = A V0, V = V0 instead of G = A, V = I finds its
// columns almost orthogonal and ends after two or three sweeps instead of eleven.  One workgroup per (run, vector):
// the unit vector from the previous `axes` column (axes = V sqrt(lambda) times the enlargement: normalising the
// column gives V back), then its image under the new covariance (symmetric: column reads are coalesced).  A run
// without a previous bound, or with a degenerate column, is flagged cold.
__global__ void __launch_bounds__(256)
    wide_warm_kernel(int D, const double* __restrict__ cov_all, const double* __restrict__ axes_all,
                     const int32_t* __restrict__ nells_prev, const int* __restrict__ active, double* V0t_all,
                     double* Wt_all, int* cold) {
  __shared__ double sv[kWideMaxD];
  __shared__ double sred[4];
  const int run = blockIdx.y, g = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (active && !active[run]) return;
  if (nells_prev[run] != 1) {
    if (t == 0) cold[run] = 1;
    return;
  }
  const double* axes = axes_all + (size_t)run * D * D;
  const double* cov = cov_all + (size_t)run * D * D;
  double ss = 0.0;
  for (int i = t; i < D; i += 256) {
    const double x = axes[(size_t)i * D + g];
    sv[i] = x;
    ss = fma(x, x, ss);
  }
  ss = wave_sum(ss);
  if (lane == 0) sred[wv] = ss;
  __syncthreads();
  const double nn = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  if (!(nn > 0.0) || !isfinite(nn)) {
    if (t == 0) cold[run] = 1;
    return;
  }
  const double inv = 1.0 / sqrt(nn);
  __syncthreads();
  for (int i = t; i < D; i += 256) {
    sv[i] *= inv;
    V0t_all[((size_t)run * D + g) * D + i] = sv[i];
  }
  __syncthreads();
  for (int i = t; i < D; i += 256) {
    double acc = 0.0;
    for (int k = 0; k < D; ++k) acc = fma(cov[(size_t)k * D + i], sv[k], acc);
    Wt_all[((size_t)run * D + g) * D + i] = acc;
  }
}

__device__ __forceinline__ bool eig_barrier(int* bar, int target, int* ok_flag) {
  // every wave's (agent-scope, write-through) stores acknowledged before thread 0 announces the arrival:
  // __syncthreads() = s_waitcnt lgkmcnt(0) + s_barrier does not wait for them (see rebuild.hip: parts_barrier)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    long long spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1ll << 20)) {  // partners never arrived (would otherwise hang the device)
        ok = 0;
        break;
      }
    }
    *ok_flag = ok;
  }
  __syncthreads();
  return *ok_flag != 0;
}

__global__ void __launch_bounds__(kRT) wide_eig_kernel(WideEigArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_rot, s_ok;
  const int D = a.D, B = a.B, b = a.b, M = 2 * B, CL = 2 * D;
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), nwv = kRT / 64;
  const int run = blockIdx.x / B, w = blockIdx.x - run * B;
  if (a.active && !a.active[run]) return;  // all B workgroups of the run leave together: no barrier is missed
  double* col = (double*)smem;  // 2b columns x CL
  const double* cov = a.cov + (size_t)run * D * D;
  double* xb = a.xbuf + (size_t)run * 2 * M * b * CL;
  int* bar = a.bar + run;
  int* rot = a.rot + (size_t)run * kEigMaxSweeps;
  const double tol2 = (double)D * (2.220446049250313e-16 * 2.220446049250313e-16);
  int nbar = 0;
  long long tp_ = (a.dbg ? clock64() : 0ll), cy_rot = 0, cy_st = 0, cy_bar = 0, cy_ld = 0;

  // circle method: the pair of blocks workgroup w holds in round r
  auto pair_of = [&](int r, int& top, int& bot) {
    if (w == 0) {
      top = M - 1;
      bot = r % (M - 1);
    } else {
      top = (r + w) % (M - 1);
      bot = (r - w + 2 * (M - 1)) % (M - 1);
    }
  };
  // one rotation: columns p, q of the LDS tile (wave-uniform p, q)
  auto rotate = [&](int p, int q) -> int {
    double* P = col + (size_t)p * CL;
    double* Q = col + (size_t)q * CL;
    double al = 0.0, be = 0.0, ga = 0.0;
    for (int i = lane; i < D; i += 64) {
      const double x = P[i], y = Q[i];
      al = fma(x, x, al);
      be = fma(y, y, be);
      ga = fma(x, y, ga);
    }
    al = wave_sum(al);
    be = wave_sum(be);
    ga = wave_sum(ga);
    if (!(ga * ga > tol2 * (al * be))) return 0;  // orthogonal already (or a zero / padding column)
    double c, sn;  // the 2x2 problem [[al, ga], [ga, be]]: same rotation as the two-sided form
    dh_eig::jacobi_rotation(al, be, ga, c, sn);
    for (int i = lane; i < CL; i += 64) {
      const double x = P[i], y = Q[i];
      P[i] = c * x - sn * y;
      Q[i] = sn * x + c * y;
    }
    return 1;
  };

  // scale to max |cov_ij| = 1 (the rotation formula squares squared column norms)
  __shared__ double s_red[kRT / 64];
  double amax = 0.0;
  for (int e = t; e < D * D; e += kRT) amax = fmax(amax, fabs(cov[e]));
  amax = block_max_1024(amax, s_red);
  const double scale = (amax > 0.0 && isfinite(amax)) ? 1.0 / amax : 1.0;
  // initial blocks: G column g = column g of the covariance, V column = e_g; padding columns zero
  {
    int top, bot;
    pair_of(0, top, bot);
    for (int e = t; e < 2 * b * CL; e += kRT) {
      const int c = e / CL, i = e - c * CL;
      const int g = (c < b ? top : bot) * b + (c < b ? c : c - b);
      double v = 0.0;
      if (g < D) v = i < D ? cov[(size_t)g * D + i] * scale : (i - D == g ? 1.0 : 0.0);
      col[e] = v;
    }
    if (t == 0) s_rot = 0;
    __syncthreads();
  }
  bool ok = true, converged = false;
  int parity = 0, sweep = 0;
  for (sweep = 0; sweep < kEigMaxSweeps && ok && !converged; ++sweep) {
    for (int r = 0; r < M - 1 && ok; ++r) {
      int nrot = 0;
      long long c0_ = (a.dbg ? clock64() : 0ll);
      if (r == 0) {
        // all pairs among the 2b columns: circle method inside the tile (2b - 1 steps of b pairs)
        const int m2 = 2 * b;
        for (int st = 0; st < m2 - 1; ++st) {
          for (int k = wv; k < b; k += nwv) {
            int p, q;
            if (k == 0) {
              p = m2 - 1;
              q = st;
            } else {
              p = (st + k) % (m2 - 1);
              q = (st - k + 2 * (m2 - 1)) % (m2 - 1);
            }
            nrot += rotate(p, q);
          }
          __syncthreads();
        }
      } else {
        // pairs across the two blocks: step st pairs column i with column b + (i + st) % b
        for (int st = 0; st < b; ++st) {
          for (int k = wv; k < b; k += nwv) nrot += rotate(k, b + (k + st) % b);
          __syncthreads();
        }
      }
      if (lane == 0 && nrot) atomicAdd(&s_rot, nrot);
      __syncthreads();
      cy_rot += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      const bool last_round = r == M - 2;
      if (M > 2) {
        // hand the blocks on
        int top, bot;
        pair_of(r, top, bot);
        double* dst = xb + (size_t)parity * M * b * CL;
        double* d_top = dst + (size_t)top * b * CL;
        double* d_bot = dst + (size_t)bot * b * CL - (size_t)b * CL;
        const int half = b * CL;
        for (int e = t; e < 2 * half; e += kRT)
          __hip_atomic_store((e < half ? d_top : d_bot) + e, col[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (last_round && t == 0) {
        if (s_rot) __hip_atomic_fetch_add(rot + sweep, s_rot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_rot = 0;
      }
      __syncthreads();  // (stores drained)
      cy_st += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      if (B > 1) {
        ++nbar;
        ok = eig_barrier(bar, B * nbar, &s_ok);
      } else {
        __syncthreads();
      }
      cy_bar += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      if (last_round) {
        const int total = __hip_atomic_load(rot + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        converged = total == 0;
      }
      if (M > 2 && ok && !(last_round && converged)) {
        int top, bot;
        pair_of((r + 1) % (M - 1), top, bot);
        const double* src = xb + (size_t)parity * M * b * CL;
        const double* s_top = src + (size_t)top * b * CL;
        const double* s_bot = src + (size_t)bot * b * CL - (size_t)b * CL;
        const int half = b * CL, tot = 2 * half;
        for (int e0 = t; e0 < tot; e0 += 8 * kRT) {  // eight requests in flight per thread
          double tmp[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * kRT;
            tmp[q] = e < tot ? __hip_atomic_load((e < half ? s_top : s_bot) + e, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                             : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * kRT;
            if (e < tot) col[e] = tmp[q];
          }
        }
        parity ^= 1;
        __syncthreads();
      }
      cy_ld += (a.dbg ? clock64() : 0ll) - c0_;
    }
  }
  if (a.dbg && t == 0 && blockIdx.x == 0)
    printf("wide_eig: D %d, %d workgroups x %d columns, %d sweeps, %lld cycles (rotations %lld, hand-on %lld, barrier %lld, pick-up %lld)\n",
           D, B, 2 * b, sweep, (long long)((a.dbg ? clock64() : 0ll) - tp_), cy_rot, cy_st, cy_bar, cy_ld);
  // results: the blocks this workgroup holds now (those of the last round it worked on)
  if (ok && converged) {
    int top, bot;
    pair_of(M - 2, top, bot);
    double* lam = a.lam + (size_t)run * D;
    double* V = a.V + (size_t)run * D * D;
    for (int c = wv; c < 2 * b; c += nwv) {
      const int g = (c < b ? top : bot) * b + (c < b ? c : c - b);
      if (g >= D) continue;
      const double* P = col + (size_t)c * CL;
      double dot = 0.0;
      for (int i = lane; i < D; i += 64) dot = fma(P[D + i], P[i], dot);
      dot = wave_sum(dot);
      if (lane == 0) lam[g] = dot * amax;
      for (int i = lane; i < D; i += 64) V[(size_t)i * D + g] = P[D + i];
    }
  }
  if (t == 0 && w == 0) a.ok[run] = (ok && converged) ? 1 : 0;
}

constexpr int kGS = 33;  // row stride of the 32 x 32 Gram / rotation matrices
__global__ void __launch_bounds__(kRT) wide_eig2_kernel(WideEigArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_rot, s_ok;
  const int D = a.D, B = a.B, b = a.b, M = 2 * B, CL = 2 * D + 1;  // CL odd: conflict-free writes across columns
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), nwv = kRT / 64;
  const int run = blockIdx.x / B, w = blockIdx.x - run * B;
  if (a.active && !a.active[run]) return;  // all B workgroups of the run leave together: no barrier is missed
  double* col = (double*)smem;              // 2b columns x CL: [G (D) | V (D) | pad]
  double* gm0 = col + (size_t)2 * b * CL;   // Gram matrix of the G parts, two buffers of 32 x kGS
  double* jm0 = gm0 + 2 * 32 * kGS;         // accumulated rotations of the round, two buffers
  double* gpart = jm0 + 2 * 32 * kGS;       // 12 partial Gram tiles of 256
  double* coefa = gpart + 12 * 256;         // per column: new = coefa * own + coefb * partner
  double* coefb = coefa + 32;
  int* partner = (int*)(coefb + 32);        // 32
  const double* cov = a.cov + (size_t)run * D * D;
  double* xb = a.xbuf + (size_t)run * 2 * M * b * CL;
  int* bar = a.bar + run;
  int* rot = a.rot + (size_t)run * kEigMaxSweeps;
  const double tol2 = (double)D * (2.220446049250313e-16 * 2.220446049250313e-16);
  int nbar = 0;
  long long tp_ = (a.dbg ? clock64() : 0ll), cy_rot = 0, cy_st = 0, cy_bar = 0, cy_ld = 0, cy_a = 0, cy_b = 0, cy_c = 0;

  // circle method: the pair of blocks workgroup w holds in round r
  auto pair_of = [&](int r, int& top, int& bot) {
    if (w == 0) {
      top = M - 1;
      bot = r % (M - 1);
    } else {
      top = (r + w) % (M - 1);
      bot = (r - w + 2 * (M - 1)) % (M - 1);
    }
  };
  // scale to max |cov_ij| = 1 (the rotation formula squares squared column norms)
  __shared__ double s_red[kRT / 64];
  double amax = 0.0;
  for (int e = t; e < D * D; e += kRT) amax = fmax(amax, fabs(cov[e]));
  amax = block_max_1024(amax, s_red);
  const double scale = (amax > 0.0 && isfinite(amax)) ? 1.0 / amax : 1.0;
  // initial blocks: G column g = column g of the covariance, V column = e_g (cold), or G = cov v0_g, V = v0_g with
  // the previous rebuild's eigenvectors v0 (warm, see wide_warm_kernel); padding columns zero
  const bool warm = a.V0t && a.Wt && a.cold && a.cold[run] == 0;
  const double* V0t = a.V0t ? a.V0t + (size_t)run * D * D : nullptr;
  const double* Wt = a.Wt ? a.Wt + (size_t)run * D * D : nullptr;
  {
    int top, bot;
    pair_of(0, top, bot);
    for (int e = t; e < 2 * b * CL; e += kRT) {
      const int c = e / CL, i = e - c * CL;
      const int g = (c < b ? top : bot) * b + (c < b ? c : c - b);
      double v = 0.0;
      if (g < D && i < 2 * D) {
        if (warm)
          v = i < D ? Wt[(size_t)g * D + i] * scale : V0t[(size_t)g * D + (i - D)];
        else
          v = i < D ? cov[(size_t)g * D + i] * scale : (i - D == g ? 1.0 : 0.0);
      }
      col[e] = v;
    }
    if (t == 0) s_rot = 0;
    __syncthreads();
  }
  bool ok = true, converged = false;
  int parity = 0, sweep = 0;
  for (sweep = 0; sweep < kEigMaxSweeps && ok && !converged; ++sweep) {
    for (int r = 0; r < M - 1 && ok; ++r) {
      int nrot = 0;
      long long c0_ = (a.dbg ? clock64() : 0ll);
      const int m2 = 2 * b, nt = (m2 + 15) >> 4;
      // (a) Gram matrix of the G parts on the matrix cores: tiles (0,0) (0,1) (1,1) x four slices of
      // the rows, partial tiles summed in slice order
      if (wv < 12) {
        const int tt = wv % 3, kq = wv / 3;
        const int ib = tt == 2 ? 1 : 0, jb = tt == 0 ? 0 : 1;
        wacc acc = {0.0, 0.0, 0.0, 0.0};
        if (jb < nt) {
          const int ca = ib * 16 + (lane & 15), cb = jb * 16 + (lane & 15);
          const int ksteps = (D + 3) >> 2, per = (ksteps + 3) >> 2;
          const int k1 = min(ksteps, (kq + 1) * per);
          for (int ks = kq * per; ks < k1; ++ks) {
            const int i = 4 * ks + (lane >> 4);
            const double fa = (ca < m2 && i < D) ? col[(size_t)ca * CL + i] : 0.0;
            const double fb = (cb < m2 && i < D) ? col[(size_t)cb * CL + i] : 0.0;
            acc = W_MFMA(fa, fb, acc);
          }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) gpart[(kq * 3 + tt) * 256 + ((lane >> 4) + 4 * q4) * 16 + (lane & 15)] = acc[q4];
      }
      __syncthreads();
      if (t < 3 * 256) {
        const int tt = t >> 8, e = t & 255, ib = tt == 2 ? 1 : 0, jb = tt == 0 ? 0 : 1;
        const double v = ((gpart[tt * 256 + e] + gpart[(3 + tt) * 256 + e]) + gpart[(6 + tt) * 256 + e]) +
                         gpart[(9 + tt) * 256 + e];
        const int i = ib * 16 + (e >> 4), j = jb * 16 + (e & 15);
        gm0[i * kGS + j] = v;
        if (tt == 1) gm0[j * kGS + i] = v;
      }
      {
        const int i = t >> 5, j = t & 31;
        jm0[i * kGS + j] = i == j ? 1.0 : 0.0;
      }
      __syncthreads();
      cy_a += (a.dbg ? clock64() : 0ll) - c0_;
      long long c1_ = (a.dbg ? clock64() : 0ll);
      // (b) the round's rotations on the small matrices: angles from the Gram matrix, Gm <- R^T Gm R,
      // J <- J R (element-wise with the partner map: new column = coefa * own + coefb * partner)
      int cur = 0;
      const int nsteps = r == 0 ? m2 - 1 : b;
      for (int st = 0; st < nsteps; ++st) {
        const double* Gs = gm0 + cur * 32 * kGS;
        const double* Js = jm0 + cur * 32 * kGS;
        double* Gd = gm0 + (cur ^ 1) * 32 * kGS;
        double* Jd = jm0 + (cur ^ 1) * 32 * kGS;
        if (t < b) {
          int p, q;
          if (r == 0) {
            if (t == 0) {
              p = m2 - 1;
              q = st;
            } else {
              p = (st + t) % (m2 - 1);
              q = (st - t + 2 * (m2 - 1)) % (m2 - 1);
            }
          } else {
            p = t;
            q = b + (t + st) % b;
          }
          const double al = Gs[p * kGS + p], be = Gs[q * kGS + q], ga = Gs[p * kGS + q];
          double c = 1.0, sn = 0.0;
          if (ga * ga > tol2 * (al * be)) {  // else orthogonal already (or a zero / padding column)
            dh_eig::jacobi_rotation(al, be, ga, c, sn);
            ++nrot;
          }
          partner[p] = q;
          partner[q] = p;
          coefa[p] = c;
          coefb[p] = -sn;
          coefa[q] = c;
          coefb[q] = sn;
        }
        __syncthreads();
        {
          const int i = t >> 5, j = t & 31;
          if (i < m2 && j < m2) {
            const int pi = partner[i], pj = partner[j];
            const double ai = coefa[i], bi = coefb[i], aj = coefa[j], bj = coefb[j];
            const double g00 = Gs[i * kGS + j], g01 = Gs[i * kGS + pj], g10 = Gs[pi * kGS + j],
                         g11 = Gs[pi * kGS + pj];
            Gd[i * kGS + j] = ai * (aj * g00 + bj * g01) + bi * (aj * g10 + bj * g11);
            Jd[i * kGS + j] = aj * Js[i * kGS + j] + bj * Js[i * kGS + pj];
          }
        }
        __syncthreads();
        cur ^= 1;
      }
      cy_b += (a.dbg ? clock64() : 0ll) - c1_;
      c1_ = (a.dbg ? clock64() : 0ll);
      // (c) the columns (G and V parts) times the accumulated rotation, on the matrix cores; a wave
      // owns 16 rows at a time: all of their operands are in registers before the first write
      {
        const double* Jf = jm0 + cur * 32 * kGS;
        const int lj = lane & 15, lk = lane >> 4;
        for (int rt = wv; rt * 16 < CL; rt += nwv) {
          const int i = rt * 16 + lj;
          double fa[8];
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const int kc = 4 * ks + lk;
            fa[ks] = (kc < m2 && i < CL) ? col[(size_t)kc * CL + i] : 0.0;
          }
          wacc acc[2];
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {
            acc[jb] = (wacc){0.0, 0.0, 0.0, 0.0};
            if (jb < nt) {
#pragma unroll
              for (int ks = 0; ks < 8; ++ks) {
                const int kc = 4 * ks + lk, jc = jb * 16 + lj;
                const double fb = (kc < m2 && jc < m2) ? Jf[kc * kGS + jc] : 0.0;
                acc[jb] = W_MFMA(fa[ks], fb, acc[jb]);
              }
            }
          }
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {
            const int jc = jb * 16 + lj;
            if (jb < nt && jc < m2) {
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const int row = rt * 16 + lk + 4 * q4;
                if (row < CL) col[(size_t)jc * CL + row] = acc[jb][q4];
              }
            }
          }
        }
      }
      __syncthreads();
      cy_c += (a.dbg ? clock64() : 0ll) - c1_;
      if (nrot) atomicAdd(&s_rot, nrot);
      __syncthreads();
      cy_rot += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      const bool last_round = r == M - 2;
      if (M > 2) {
        // hand the blocks on
        int top, bot;
        pair_of(r, top, bot);
        double* dst = xb + (size_t)parity * M * b * CL;
        double* d_top = dst + (size_t)top * b * CL;
        double* d_bot = dst + (size_t)bot * b * CL - (size_t)b * CL;
        const int half = b * CL;
        for (int e = t; e < 2 * half; e += kRT)
          __hip_atomic_store((e < half ? d_top : d_bot) + e, col[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (last_round && t == 0) {
        if (s_rot) __hip_atomic_fetch_add(rot + sweep, s_rot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_rot = 0;
      }
      __syncthreads();  // (stores drained)
      cy_st += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      if (B > 1) {
        ++nbar;
        ok = eig_barrier(bar, B * nbar, &s_ok);
      } else {
        __syncthreads();
      }
      cy_bar += (a.dbg ? clock64() : 0ll) - c0_;
      c0_ = (a.dbg ? clock64() : 0ll);
      if (last_round) {
        const int total = __hip_atomic_load(rot + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        converged = total == 0;
      }
      if (M > 2 && ok && !(last_round && converged)) {
        int top, bot;
        pair_of((r + 1) % (M - 1), top, bot);
        const double* src = xb + (size_t)parity * M * b * CL;
        const double* s_top = src + (size_t)top * b * CL;
        const double* s_bot = src + (size_t)bot * b * CL - (size_t)b * CL;
        const int half = b * CL, tot = 2 * half;
        for (int e0 = t; e0 < tot; e0 += 8 * kRT) {  // eight requests in flight per thread
          double tmp[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * kRT;
            tmp[q] = e < tot ? __hip_atomic_load((e < half ? s_top : s_bot) + e, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                             : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * kRT;
            if (e < tot) col[e] = tmp[q];
          }
        }
        parity ^= 1;
        __syncthreads();
      }
      cy_ld += (a.dbg ? clock64() : 0ll) - c0_;
    }
  }
  if (a.dbg && t == 0 && blockIdx.x == 0)
    printf("wide_eig2: D %d, %d workgroups x %d columns, %d sweeps, %lld cycles (rotations %lld = Gram %lld + steps %lld + apply %lld, hand-on %lld, barrier %lld, pick-up %lld)\n",
           D, B, 2 * b, sweep, (long long)((a.dbg ? clock64() : 0ll) - tp_), cy_rot, cy_a, cy_b, cy_c, cy_st, cy_bar, cy_ld);
  // results: the blocks this workgroup holds now (those of the last round it worked on)
  if (ok && converged) {
    int top, bot;
    pair_of(M - 2, top, bot);
    double* lam = a.lam + (size_t)run * D;
    double* V = a.V + (size_t)run * D * D;
    for (int c = wv; c < 2 * b; c += nwv) {
      const int g = (c < b ? top : bot) * b + (c < b ? c : c - b);
      if (g >= D) continue;
      const double* P = col + (size_t)c * CL;
      double dot = 0.0;
      for (int i = lane; i < D; i += 64) dot = fma(P[D + i], P[i], dot);
      dot = wave_sum(dot);
      if (lane == 0) lam[g] = dot * amax;
      for (int i = lane; i < D; i += 64) V[(size_t)i * D + g] = P[D + i];
    }
  }
  if (t == 0 && w == 0) a.ok[run] = (ok && converged) ? 1 : 0;
}

__global__ void __launch_bounds__(kRT) wide_single_kernel(WideRebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = a.d, n = a.n, t = threadIdx.x, run = blockIdx.x;
  if (a.active && !a.active[run]) return;
  const int LD = D | 1;
  double* tile = (double*)smem;            // tp x LD
  double* mean = tile + (size_t)a.tp * LD;  // D
  double* lam = mean + D;                  // D
  double* red = lam + D;                   // 64
  double* rc = red + 64;                   // D/2+1
  double* rs = rc + (kWideMaxD / 2 + 1);
  double* part = rs + (kWideMaxD / 2 + 1);     // ceil(D/16) x 64 partial quadratic forms (fmax)
  int* rp = (int*)(part + (size_t)((D + 15) / 16) * 64);  // 2*(kWideMaxD/2+1)
  int* order = rp + 2 * (kWideMaxD / 2 + 1);   // D
  const double* pts = a.pts + (size_t)run * n * D;
  double* A = a.wsA + (size_t)run * D * D;
  double* V = a.wsV + (size_t)run * D * D;
  double* cov = a.wscov + (size_t)run * D * D;
  double* o_am = a.ams + (size_t)run * D * D;
  double* o_ax = a.axes + (size_t)run * D * D;
  int status = DH_OK;
  if (n <= 1) status = DH_ERR_VALUE;
  if (a.phase == 2 && a.status[run] != DH_OK) return;  // the covariance kernels already said why
  long long tp_ = clock64();
#define WPH(name)                                                                   \
  do {                                                                              \
    if (a.dbg && t == 0 && run == 0) {                                              \
      const long long now_ = clock64();                                             \
      printf("wide_single %s %lld\n", name, (long long)(now_ - tp_)); \
      tp_ = now_;                                                                   \
    }                                                                               \
  } while (0)

  if (status == DH_OK && a.phase == 2) {
    for (int k = t; k < D; k += kRT) mean[k] = a.ctrs[(size_t)run * D + k];
    __syncthreads();
  }
  if (status == DH_OK && a.phase != 2) {
    // ---- mean (np.mean axis 0) ----
    {
      // thread (j, g): dims j = t % DP..., use simple map: each thread sums a strided set of points
      const int G = kRT / D > 0 ? kRT / D : 1;
      const int j = t % D, g = t / D;
      double acc = 0.0;
      if (g < G)
        for (int p = g; p < n; p += G) acc += pts[(size_t)p * D + j];
      // reduce the G partials per dimension through global scratch (A reused)
      if (g < G) A[(size_t)g * D + j] = acc;
      __threadfence_block();
      __syncthreads();
      if (t < D) {
        double s = 0.0;
        for (int gg = 0; gg < G; ++gg) s += A[(size_t)gg * D + t];
        mean[t] = s / (double)n;
      }
      __syncthreads();
    }
    WPH("mean");
    wide_cov_range(pts, D, a.tp, 0, n, mean, tile, cov, 1.0 / (double)(n - 1), true);
    WPH("cov");
  }
  if (status == DH_OK) {
    // ---- improve_covar_mat + fmax passes (bounding.py:1311-1384, 1423-1457) ----
    const double lim = 1.0 - 1e-3;
    for (int pass = 0; pass < 2 && status == DH_OK; ++pass) {
      bool good = false;
      int failed = 0, trial = 0;
      for (trial = 0; trial < 100; ++trial) {
        failed = 0;
        const bool pre = a.phase == 2 && pass == 0 && trial == 0 && a.eig_ok[run] == 1;
        bool fin = true;
        if (!pre) {
          for (int e = t; e < D * D; e += kRT) A[e] = cov[e];
          __threadfence_block();
          __syncthreads();
          WPH("copy");
          fin = jacobi_global(A, V, a.wsW + (size_t)run * 4 * (size_t)((D + 1) & ~1) * ((D + 1) & ~1), D, rc, rs, red);
          WPH("jacobi");
        }
        double top = -INFINITY, bot = INFINITY;
        bool allfin = fin;
        if (fin) {
          for (int k = t; k < D; k += kRT) lam[k] = pre ? a.lam_pre[(size_t)run * D + k] : A[(size_t)k * D + k];
          __syncthreads();
          for (int k = 0; k < D; ++k) {
            const double l = lam[k];
            if (!isfinite(l)) allfin = false;
            top = fmax(top, l);
            bot = fmin(bot, l);
          }
        }
        if (allfin) {
          if (top <= 0.0)
            failed = 2;
          else if (bot < top / 1e12)
            failed = 1;
          else
            break;
        } else {
          failed = 2;
        }
        if (failed == 1) {
          __syncthreads();
          if (t < D) lam[t] = fmax(lam[t], 10.0 * top / 1e12);
          __syncthreads();
          for (int e = t; e < D * D; e += kRT) {
            const int i = e / D, j = e - i * D;
            double s = 0.0;
            for (int k = 0; k < D; ++k) s = fma(V[(size_t)i * D + k] * lam[k], V[(size_t)j * D + k], s);
            cov[e] = s;
          }
        } else {
          const double coeff = 1e-10 * pow(1e10, (double)trial / 99.0);
          for (int e = t; e < D * D; e += kRT) {
            const int i = e / D, j = e - i * D;
            cov[e] = (1.0 - coeff) * cov[e] + coeff * (i == j ? 1.0 : 0.0);
          }
        }
        __threadfence_block();
        __syncthreads();
      }
      if (failed > 0) {
        for (int e = t; e < D * D; e += kRT) {
          const int i = e / D, j = e - i * D;
          const double v = (i == j) ? 1.0 : 0.0;
          cov[e] = v;
          V[e] = v;
        }
        if (t < D) lam[t] = 1.0;
        __threadfence_block();
        __syncthreads();
      } else {
        good = trial == 0;
      }
      WPH("regularize");
      // sort ascending + canonical signs: order[] by rank; AX/AM from (V, lam)
      for (int k = t; k < D; k += kRT) {
        const double mine = lam[k];
        int rank = 0;
        for (int j = 0; j < D; ++j) {
          const double o = lam[j];
          if (o < mine || (o == mine && j < k)) ++rank;
        }
        order[rank] = k;
      }
      __syncthreads();
      // sign of each sorted eigenvector (largest |component| positive) -> rc[] reused as sign
      for (int k = t; k < D; k += kRT) {
        const int src = order[k];
        double best = 0.0, val = 1.0;
        for (int i = 0; i < D; ++i) {
          const double vv = V[(size_t)i * D + src];
          if (fabs(vv) > best) {
            best = fabs(vv);
            val = vv;
          }
        }
        // store the sign in the A work matrix diagonal slot k (A is dead here)
        A[(size_t)k * D + k] = val < 0.0 ? -1.0 : 1.0;
      }
      __threadfence_block();
      __syncthreads();
      for (int e = t; e < D * D; e += kRT) {
        const int i = e / D, k = e - i * D;
        const int src = order[k];
        o_ax[e] = A[(size_t)k * D + k] * V[(size_t)i * D + src] * sqrt(lam[src]);
      }
      // am = (V / lam) V^T on the matrix cores: block (ib <= jb) per wave step, K = D in steps of 4
      {
        const int lane = t & 63, wv = t >> 6, lj = lane & 15, lk = lane >> 4;
        const int nbk = (D + 15) >> 4, npairs = nbk * (nbk + 1) / 2, ksteps = (D + 3) >> 2;
        for (int pr = wv; pr < npairs; pr += kRT / 64) {
          int ib = 0, rem = pr;
          while (rem >= nbk - ib) {
            rem -= nbk - ib;
            ++ib;
          }
          const int jb = ib + rem;
          const int ia = ib * 16 + lj, ja = jb * 16 + lj;
          wacc acc = {0.0, 0.0, 0.0, 0.0};
          for (int ks = 0; ks < ksteps; ++ks) {
            const int k = ks * 4 + lk;
            const bool kv = k < D;
            const double fa = (kv && ia < D) ? V[(size_t)ia * D + k] / lam[k] : 0.0;
            const double fb = (kv && ja < D) ? V[(size_t)ja * D + k] : 0.0;
            acc = W_MFMA(fa, fb, acc);
          }
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int i = ib * 16 + lk + 4 * q4, j = jb * 16 + lj;
            if (i < D && j < D) {
              o_am[(size_t)i * D + j] = acc[q4];
              o_am[(size_t)j * D + i] = acc[q4];
            }
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      WPH("sort+ax+am");
      if (a.phase == 2 && a.split_tail && pass == 0 && good) {
        for (int k = t; k < D; k += kRT) {
          a.lam_ws[(size_t)run * D + k] = lam[k];
          a.order_ws[(size_t)run * D + k] = order[k];
        }
        if (t == 0) a.fast[run] = 1;
        return;
      }
      const double best = wide_fmax_range(pts, D, a.tp, 0, n, mean, tile, part, o_am);
      const double fmx = block_max_1024(best, red);
      WPH("fmax");
      if (pass == 0 && fmx > lim) {
        const double mult = fmx / lim, rt = sqrt(mult);
        for (int e = t; e < D * D; e += kRT) {
          cov[e] *= mult;
          o_am[e] /= mult;
          o_ax[e] *= rt;
        }
        __syncthreads();
        if (t < D) lam[t] *= mult;
        __threadfence_block();
        __syncthreads();
      }
      if (pass == 1 && fmx >= 1.0) status = DH_ERR_CONTAIN;
      if (good) break;
    }
  }
  if (status == DH_OK) {
    bool ok = true;
    double slog = 0.0;
    for (int k = 0; k < D; ++k) {
      const double l = lam[k];
      if (!(l > 0.0) || !isfinite(l)) ok = false;
      slog += log(l);
    }
    if (!ok)
      status = DH_ERR_VALUE;
    else {
      for (int e = t; e < D * D; e += kRT) a.covs[(size_t)run * D * D + e] = cov[e];
      for (int k = t; k < D; k += kRT) {
        a.ctrs[(size_t)run * D + k] = mean[k];
        a.axlens[(size_t)run * D + k] = sqrt(lam[order[k]]);
      }
      if (t == 0) a.logvols[run] = a.prefactor + 0.5 * slog;
    }
  }
  if (t == 0) a.status[run] = status;
}

// ---- the data-parallel parts of the rebuild over P workgroups per run -------------------------
// mean: partial sums per chunk; covariance: partial Gram matrices per chunk (mean re-derived from
// the partials in chunk order by every workgroup), reduced in chunk order; Mahalanobis maximum:
// partial maxima.  Everything is summed in a fixed order: results do not depend on scheduling.
__global__ void __launch_bounds__(kRT) wide_mean_part_kernel(WideRebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* sacc = (double*)smem;  // G x D
  const int D = a.d, t = threadIdx.x, run = blockIdx.x / a.P, q = blockIdx.x - run * a.P;
  if (a.active && !a.active[run]) return;
  const int pbeg = min(a.n, q * a.chunk), pend = min(a.n, pbeg + a.chunk);
  const double* pts = a.pts + (size_t)run * a.n * D;
  const int G = kRT / D > 0 ? kRT / D : 1;
  const int j = t % D, g = t / D;
  double acc = 0.0;
  if (g < G)
    for (int p = pbeg + g; p < pend; p += G) acc += pts[(size_t)p * D + j];
  if (g < G) sacc[(size_t)g * D + j] = acc;
  __syncthreads();
  if (t < D) {
    double sum = 0.0;
    for (int gg = 0; gg < G; ++gg) sum += sacc[(size_t)gg * D + t];
    a.meanpart[((size_t)run * a.P + q) * D + t] = sum;
  }
}

__device__ __forceinline__ void wide_mean_from_parts(const WideRebuildArgs& a, int run, double* mean) {
  for (int k = threadIdx.x; k < a.d; k += kRT) {
    double sum = 0.0;
    for (int q = 0; q < a.P; ++q) sum += a.meanpart[((size_t)run * a.P + q) * a.d + k];
    mean[k] = sum / (double)a.n;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kRT) wide_cov_part_kernel(WideRebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = a.d, LD = D | 1, run = blockIdx.x / a.P, q = blockIdx.x - run * a.P;
  if (a.active && !a.active[run]) return;
  double* tile = (double*)smem;
  double* mean = tile + (size_t)a.tp * LD;
  const int pbeg = min(a.n, q * a.chunk), pend = min(a.n, pbeg + a.chunk);
  wide_mean_from_parts(a, run, mean);
  wide_cov_range(a.pts + (size_t)run * a.n * D, D, a.tp, pbeg, pend, mean, tile,
                 a.covpart + ((size_t)run * a.P + q) * D * D, 1.0, false);
}

__global__ void __launch_bounds__(256) wide_cov_reduce_kernel(WideRebuildArgs a) {
  const int D = a.d, run = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (a.active && !a.active[run]) return;
  if (e < D * D) {
    const int i = e / D, j = e - i * D;
    const size_t src = (i >> 4) <= (j >> 4) ? (size_t)i * D + j : (size_t)j * D + i;
    double sum = 0.0;
    for (int q = 0; q < a.P; ++q) sum += a.covpart[((size_t)run * a.P + q) * D * D + src];
    a.wscov[(size_t)run * D * D + e] = sum * (1.0 / (double)(a.n - 1));
  }
  if (blockIdx.x == 0) {
    for (int k = threadIdx.x; k < D; k += 256) {
      double sum = 0.0;
      for (int q = 0; q < a.P; ++q) sum += a.meanpart[((size_t)run * a.P + q) * D + k];
      a.ctrs[(size_t)run * D + k] = sum / (double)a.n;
    }
    if (threadIdx.x == 0) a.status[run] = DH_OK;
  }
}

__global__ void __launch_bounds__(kRT) wide_fmax_part_kernel(WideRebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = a.d, LD = D | 1, t = threadIdx.x, run = blockIdx.x / a.P, q = blockIdx.x - run * a.P;
  if (!a.fast[run]) return;
  double* tile = (double*)smem;
  double* mean = tile + (size_t)a.tp * LD;
  double* red = mean + D;
  double* part = red + 64;
  const int pbeg = min(a.n, q * a.chunk), pend = min(a.n, pbeg + a.chunk);
  for (int k = t; k < D; k += kRT) mean[k] = a.ctrs[(size_t)run * D + k];
  __syncthreads();
  const double best = wide_fmax_range(a.pts + (size_t)run * a.n * D, D, a.tp, pbeg, pend, mean, tile, part,
                                      a.ams + (size_t)run * D * D);
  const double fm = block_max_1024(best, red);
  if (t == 0) a.fmaxpart[(size_t)run * a.P + q] = fm;
}

// the tail of wide_single_kernel after the Mahalanobis maximum, for the runs phase 2 left to us
__global__ void __launch_bounds__(kRT) wide_finish_kernel(WideRebuildArgs a) {
  const int D = a.d, t = threadIdx.x, run = blockIdx.x;
  if (!a.fast[run]) return;
  double fmx = -INFINITY;
  for (int q = 0; q < a.P; ++q) fmx = fmax(fmx, a.fmaxpart[(size_t)run * a.P + q]);
  const double lim = 1.0 - 1e-3;
  double* cov = a.wscov + (size_t)run * D * D;
  double* o_am = a.ams + (size_t)run * D * D;
  double* o_ax = a.axes + (size_t)run * D * D;
  const double* lam = a.lam_ws + (size_t)run * D;
  const int* order = a.order_ws + (size_t)run * D;
  const double mult = fmx > lim ? fmx / lim : 1.0, rt = sqrt(mult);
  bool ok = true;
  double slog = 0.0;
  for (int k = 0; k < D; ++k) {
    const double l = fmx > lim ? lam[k] * mult : lam[k];
    if (!(l > 0.0) || !isfinite(l)) ok = false;
    slog += log(l);
  }
  if (!ok) {
    if (t == 0) a.status[run] = DH_ERR_VALUE;
    return;
  }
  for (int e = t; e < D * D; e += kRT) {
    double c = cov[e];
    if (fmx > lim) {
      c *= mult;
      o_am[e] /= mult;
      o_ax[e] *= rt;
    }
    a.covs[(size_t)run * D * D + e] = c;
  }
  for (int k = t; k < D; k += kRT) {
    const double l = fmx > lim ? lam[order[k]] * mult : lam[order[k]];
    a.axlens[(size_t)run * D + k] = sqrt(l);
  }
  if (t == 0) {
    a.logvols[run] = a.prefactor + 0.5 * slog;
    a.status[run] = DH_OK;
  }
}

// points per LDS tile for dimension D: the largest of 64 / 32 / 16 whose tile fits
int wide_tile_points(int D) {
  const int LD = D | 1;
  const size_t other = (2 * (size_t)D + 64 + 2 * (kWideMaxD / 2 + 1) + (size_t)((D + 15) / 16) * 64) * 8 +
                       (2 * (kWideMaxD / 2 + 1) + kWideMaxD + 8) * 4;
  for (int tp = kTPMax; tp > 16; tp >>= 1)
    if ((size_t)tp * LD * 8 + other <= 159 * 1024) return tp;
  return 16;
}
size_t wide_single_lds(int D) {
  const int LD = D | 1;
  size_t dbl = (size_t)wide_tile_points(D) * LD + 2 * (size_t)D + 64 + 2 * (kWideMaxD / 2 + 1) + (size_t)((D + 15) / 16) * 64;
  size_t part = (size_t)(kRT / 64) * 64;  // fmax partials alias the tile
  if (dbl < part) dbl = part;
  return dbl * 8 + (2 * (kWideMaxD / 2 + 1) + kWideMaxD + 8) * 4;
}

int ensure_ws(dh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->rebuild_ws_cap) return DH_OK;
  if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
  if (ctx->rebuild_ws) (void)hipFree(ctx->rebuild_ws);
  ctx->rebuild_ws = nullptr;
  ctx->rebuild_ws_cap = 0;
  if (!hip_ok(ctx, hipMalloc((void**)&ctx->rebuild_ws, bytes), "hipMalloc(wide scratch)")) return DH_ERR_NOMEM;
  ctx->rebuild_ws_cap = bytes;
  return DH_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// entry points used by the dispatchers in walk.hip / walk2.hip / bound.hip /
// rebuild.hip when the dimension exceeds the register-resident limits
// ---------------------------------------------------------------------------
namespace dh {

int wide_walk_launch(dh_ctx* ctx, int kind, int problem, int k, int ndim, int ncdim, const double* u0,
                     const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                     int iters, int doubling, const int8_t* bc, const uint64_t* rng, double* u, double* v,
                     double* logl, int32_t* c0, int32_t* c1, int32_t* c2, int32_t* flags,
                     uint64_t* rng_out, const double* run_loglstar, const double* run_scale, const int* run_mode,
                     const int* run_doubling, int wpr, int my_mode, const PhiloxKey* philox) {
  WideWalkArgs a;
  a.ph = philox ? *philox : PhiloxKey{0, 0, 0};
  a.ph.offset = (a.ph.offset + 3ull) & ~3ull;  // WaveGen<RNG_PHILOX> advances in whole blocks
  if (!philox && !rng && k > 0) return fail(ctx, DH_ERR_ARG, "wide walk: no generator states");
  if (!get_problem(ctx, problem, &a.prob)) return DH_ERR_ARG;
  a.run_loglstar = run_loglstar;
  a.run_scale = run_scale;
  a.run_mode = run_mode;
  a.run_doubling = run_doubling;
  a.wpr = wpr;
  a.my_mode = my_mode;
  if (ndim > kWideMaxD) return fail(ctx, DH_ERR_ARG, "ndim=%d exceeds the wide-D limit %d", ndim, kWideMaxD);
  // transposed frames in the context scratch
  const size_t at_bytes = axes ? (size_t)m * ncdim * ncdim * 8 : 0;
  if (at_bytes > ctx->axes_t_cap) {
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    if (ctx->axes_t) (void)hipFree(ctx->axes_t);
    ctx->axes_t = nullptr;
    ctx->axes_t_cap = 0;
    if (!hip_ok(ctx, hipMalloc((void**)&ctx->axes_t, at_bytes * 2), "hipMalloc(axes_t)")) return DH_ERR_NOMEM;
    ctx->axes_t_cap = at_bytes * 2;
  }
  const size_t tot = axes ? (size_t)m * ncdim * ncdim : 0;
  if (tot)
    hipLaunchKernelGGL(wide_transpose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                     axes, m, ncdim, ctx->axes_t);
  a.k = k;
  a.ndim = ndim;
  a.ncdim = ncdim;
  a.m = m;
  a.iters = iters;
  a.kind = kind;
  a.doubling0 = doubling;
  a.scale = scale;
  a.loglstar = loglstar;
  a.u0 = u0;
  a.axes_t = ctx->axes_t;
  a.axes_idx = axes_idx;
  a.bc = bc;
  a.rng_in = rng;
  a.u = u;
  a.v = v;
  a.logl = logl;
  a.c0 = c0;
  a.c1 = c1;
  a.c2 = c2;
  a.flags = flags;
  a.rng_out = rng_out;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  a.dbg = getenv("DH_WIDE_PROF") ? 1 : 0;
  // walkers per workgroup: as many as LDS allows up to the 16 columns of one MFMA tile
  a.lds_ws = (4 * ndim) | 1;
  const size_t per_wave = (size_t)a.lds_ws * 8 + (size_t)((ndim + 1) & ~1) * 4;
  int wpw = (int)std::min<size_t>(kWalkMaxWaves, (150 * 1024) / per_wave);
  if (const char* e = getenv("DH_WIDE_WPW")) wpw = std::max(1, std::min(wpw, atoi(e)));
  wpw = std::max(1, std::min(wpw, k));
  const size_t lds = per_wave * wpw;
  static size_t lds_attr_dev[kMaxDev][8] = {};
  size_t* lds_attr = lds_attr_dev[ctx->device & (kMaxDev - 1)];
  const int kk = (kind < 0 || kind > 3 ? 3 : kind) + (philox ? 4 : 0);
  const void* fns[8] = {(const void*)wide_walk_kernel<0, RNG_PCG64>,  (const void*)wide_walk_kernel<1, RNG_PCG64>,
                        (const void*)wide_walk_kernel<2, RNG_PCG64>,  (const void*)wide_walk_kernel<3, RNG_PCG64>,
                        (const void*)wide_walk_kernel<0, RNG_PHILOX>, (const void*)wide_walk_kernel<1, RNG_PHILOX>,
                        (const void*)wide_walk_kernel<2, RNG_PHILOX>, (const void*)wide_walk_kernel<3, RNG_PHILOX>};
  if (lds > lds_attr[kk]) {
    if (!hip_ok(ctx, hipFuncSetAttribute(fns[kk], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "hipFuncSetAttribute(wide_walk)"))
      return DH_ERR_HIP;
    lds_attr[kk] = lds;
  }
  const dim3 grid((k + wpw - 1) / wpw), block(64 * wpw);
  switch (kk) {
    case 0: hipLaunchKernelGGL((wide_walk_kernel<0, RNG_PCG64>), grid, block, lds, ctx->stream, a); break;
    case 1: hipLaunchKernelGGL((wide_walk_kernel<1, RNG_PCG64>), grid, block, lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((wide_walk_kernel<2, RNG_PCG64>), grid, block, lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((wide_walk_kernel<3, RNG_PCG64>), grid, block, lds, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL((wide_walk_kernel<0, RNG_PHILOX>), grid, block, lds, ctx->stream, a); break;
    case 5: hipLaunchKernelGGL((wide_walk_kernel<1, RNG_PHILOX>), grid, block, lds, ctx->stream, a); break;
    case 6: hipLaunchKernelGGL((wide_walk_kernel<2, RNG_PHILOX>), grid, block, lds, ctx->stream, a); break;
    default: hipLaunchKernelGGL((wide_walk_kernel<3, RNG_PHILOX>), grid, block, lds, ctx->stream, a); break;
  }
  return hip_ok(ctx, hipGetLastError(), "wide walk launch") ? DH_OK : DH_ERR_HIP;
}

int wide_unif_launch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                     const double* axes, const double* ams, const double* cumprob, double loglstar,
                     const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u, double* v, double* logl,
                     int32_t* ncalls, int32_t* flags, uint64_t* rng_out, const PhiloxKey* philox,
                     const double* run_loglstar, const int* run_mode, int wpr, int my_mode, const int* run_nells,
                     int run_me) {
  WideUnifArgs a;
  a.run_loglstar = run_loglstar;
  a.run_mode = run_mode;
  a.run_nells = run_nells;
  a.wpr = wpr;
  a.my_mode = my_mode;
  a.run_me = run_mode ? run_me : 0;
  a.ph = philox ? *philox : PhiloxKey{0, 0, 0};
  a.ph.offset = (a.ph.offset + 3ull) & ~3ull;
  if (!philox && !rng && k > 0) return fail(ctx, DH_ERR_ARG, "wide unif: no generator states");
  a.propose_only = problem == -1 ? 1 : 0;
  if (a.propose_only) {
    a.prob = ProblemDev();
    a.prob.ndim = ndim;
  } else if (!get_problem(ctx, problem, &a.prob)) {
    return DH_ERR_ARG;
  }
  if (ndim > kWideMaxD) return fail(ctx, DH_ERR_ARG, "ndim=%d exceeds the wide-D limit %d", ndim, kWideMaxD);
  if (m < 0 || (m > 0 && (!ctrs || !axes)) || (m > 1 && (!ams || !cumprob)))
    return fail(ctx, DH_ERR_ARG, "unif (wide): m=%d needs centres, axes and (m > 1) precision matrices", m);
  if (run_mode && (wpr < 1 || k % wpr)) return fail(ctx, DH_ERR_ARG, "unif (wide): k=%d is not runs x %d", k, wpr);
  const size_t tot = (size_t)m * ncdim * ncdim;
  if (tot * 8 > ctx->axes_t_cap) {
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    if (ctx->axes_t) (void)hipFree(ctx->axes_t);
    ctx->axes_t = nullptr;
    ctx->axes_t_cap = 0;
    if (!hip_ok(ctx, hipMalloc((void**)&ctx->axes_t, tot * 16), "hipMalloc(axes_t)")) return DH_ERR_NOMEM;
    ctx->axes_t_cap = tot * 16;
  }
  if (tot)
    hipLaunchKernelGGL(wide_transpose_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, axes,
                       m, ncdim, ctx->axes_t);
  a.k = k;
  a.ndim = ndim;
  a.ncdim = ncdim;
  a.m = m;
  a.loglstar = loglstar;
  a.ctrs = ctrs;
  a.axes_t = ctx->axes_t;
  a.ams = ams;
  a.cumprob = cumprob;
  a.bc = bc;
  a.rng_in = rng;
  a.max_tries = max_tries > 0 ? max_tries : ((int64_t)1 << 32);
  a.u = u;
  a.v = v;
  a.logl = logl;
  a.ncalls = ncalls;
  a.flags = flags;
  a.rng_out = rng_out;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  if (philox)
    hipLaunchKernelGGL(wide_unif_kernel<RNG_PHILOX>, dim3(k), dim3(64), (size_t)(ncdim + 2 * ndim) * 8, ctx->stream, a);
  else
    hipLaunchKernelGGL(wide_unif_kernel<RNG_PCG64>, dim3(k), dim3(64), (size_t)(ncdim + 2 * ndim) * 8, ctx->stream, a);
  return hip_ok(ctx, hipGetLastError(), "wide unif launch") ? DH_OK : DH_ERR_HIP;
}

int wide_eval_launch(dh_ctx* ctx, const ProblemDev& p, int k, const double* u, double* v, double* logl) {
  if (p.ndim > kWideMaxD) return fail(ctx, DH_ERR_ARG, "ndim=%d exceeds the wide-D limit %d", p.ndim, kWideMaxD);
  hipLaunchKernelGGL(wide_eval_kernel, dim3(k), dim3(64), (size_t)2 * p.ndim * 8, ctx->stream, p, k, u, v,
                     logl);
  return hip_ok(ctx, hipGetLastError(), "wide eval launch") ? DH_OK : DH_ERR_HIP;
}

int wide_contains_launch(dh_ctx* ctx, const double* x, int k, int d, const double* ctrs, const double* ams,
                         int m, int mode, int32_t* count, uint64_t* mask, double* quad) {
  if (d > 4096) return fail(ctx, DH_ERR_ARG, "contains: d=%d too large", d);
  if (mask) {
    const size_t nw = (size_t)(k + 63) / 64 * m;
    if (!hip_ok(ctx, hipMemsetAsync(mask, 0, nw * 8, ctx->stream), "memset(mask)")) return DH_ERR_HIP;
  }
  hipLaunchKernelGGL(wide_contains_kernel, dim3(k), dim3(256), (size_t)(d + 256) * 8, ctx->stream, x, k, d,
                     ctrs, ams, m, mode, count, mask, quad);
  return hip_ok(ctx, hipGetLastError(), "wide contains launch") ? DH_OK : DH_ERR_HIP;
}

// warm_nells != null: axes / axlens still hold the previous rebuild's result for every run with warm_nells[run] == 1,
// and the eigensolver starts from it (the device-resident loop: consecutive rebuilds of the same runs)
static int wide_single_enqueue(dh_ctx* ctx, int runs, const double* pts, int n, int d, const int* active, int32_t* status,
                               double* ctrs, double* covs, double* ams, double* axes, double* axlens,
                               double* logvols, const int32_t* warm_nells = nullptr) {
  if (d > kWideMaxD) return fail(ctx, DH_ERR_ARG, "rebuild: d=%d exceeds the wide-D limit %d", d, kWideMaxD);
  const size_t lds = wide_single_lds(d);
  if (lds > 159 * 1024) return fail(ctx, DH_ERR_ARG, "rebuild: d=%d needs %zu B of LDS", d, lds);
  const size_t dd = (size_t)d * d * 8;
  const size_t pw = (size_t)((d + 1) & ~1);
  const size_t ww = 4 * pw * pw * 8;  // double-buffered Jacobi work per run
  // multi-workgroup eigensolver: blocks of b <= 16 columns, two per workgroup
  // (form 2, the default: the round's rotations on the Gram matrix, applied to the columns by one
  // product -- column stride 2d + 1 and 59 KB of small matrices beside the columns; DH_WIDE_EIG=1
  // selects the pairwise form)
  const char* e_form = getenv("DH_WIDE_EIG");
  const bool gram = !(e_form && atoi(e_form) == 1);
  const size_t clen = gram ? 2 * (size_t)d + 1 : 2 * (size_t)d;
  const size_t eig_other = gram ? ((size_t)4 * 32 * kGS + 12 * 256 + 64) * 8 + 128 : 0;
  const int bmax = std::max(1, std::min(16, (int)(((gram ? 159 : 144) * 1024 - eig_other) / (2 * clen * 8))));
  const int M = 2 * ((d + 2 * bmax - 1) / (2 * bmax)), B = M / 2, b = (d + M - 1) / M;
  const size_t xb = (size_t)2 * M * b * clen * 8;
  const size_t eig_lds = (size_t)2 * b * clen * 8 + eig_other;
  const int n_cu = ctx->num_cu;  // (the context's own device: no process-global memo)
  const char* e_eig = getenv("DH_WIDE_EIG");
  const bool split = !(e_eig && atoi(e_eig) == 0) && (long long)B * runs <= n_cu && n > 1;
  // chunks of whole 64-point tiles for the data-parallel kernels
  const int tp = wide_tile_points(d);
  const int ntiles = (n + tp - 1) / tp;
  const int ct = std::max(1, (ntiles + 31) / 32), P = (ntiles + ct - 1) / ct, chunk = ct * tp;
  const size_t part_bytes = ((size_t)P * d + (size_t)P * d * d + P + d) * 8 + (size_t)d * 4 + 64;
  const size_t ints = (size_t)(kEigMaxSweeps + 4) * 4;
  int rc = ensure_ws(ctx, (5 * dd + ww + xb + (size_t)d * 8 + ints + part_bytes) * runs + 4096);
  if (rc) return rc;
  WideRebuildArgs a;
  a.pts = pts;
  a.n = n;
  a.d = d;
  a.runs = runs;
  a.prefactor = d * log(2.0) + d * lgamma(1.5) - lgamma(d / 2.0 + 1.0);
  a.wsA = (double*)ctx->rebuild_ws;
  a.wsV = a.wsA + (size_t)runs * d * d;
  a.wscov = a.wsV + (size_t)runs * d * d;
  double* ws_v0t = a.wscov + (size_t)runs * d * d;   // warm start: previous eigenvectors (rows) ...
  double* ws_wt = ws_v0t + (size_t)runs * d * d;     // ... and their images under the new covariance
  a.wsW = ws_wt + (size_t)runs * d * d;
  double* x_buf = a.wsW + (size_t)runs * 4 * pw * pw;
  double* lam_pre = x_buf + (size_t)runs * 2 * M * b * clen;
  a.meanpart = lam_pre + (size_t)runs * d;
  a.covpart = a.meanpart + (size_t)runs * P * d;
  a.fmaxpart = a.covpart + (size_t)runs * P * d * d;
  a.lam_ws = a.fmaxpart + (size_t)runs * P;
  a.order_ws = (int*)(a.lam_ws + (size_t)runs * d);
  // bar[runs] | rot[runs x kEigMaxSweeps] | ok[runs] | fast[runs] | cold[runs]
  int* eig_int = a.order_ws + (size_t)runs * d + 2;
  int* eig_cold = eig_int + (size_t)runs * (3 + kEigMaxSweeps);
  a.dbg = getenv("DH_WIDE_PROF") ? 1 : 0;
  a.phase = 0;
  a.lam_pre = lam_pre;
  a.eig_ok = eig_int + (size_t)runs * (1 + kEigMaxSweeps);
  a.fast = eig_int + (size_t)runs * (2 + kEigMaxSweeps);
  a.split_tail = 0;
  a.tp = tp;
  a.P = P;
  a.chunk = chunk;
  a.status = status;
  a.ctrs = ctrs;
  a.covs = covs;
  a.ams = ams;
  a.axes = axes;
  a.axlens = axlens;
  a.logvols = logvols;
  a.active = active;
  const int LD = d | 1;
  const size_t lds_part = ((size_t)tp * LD + d + 64 + (size_t)((d + 15) / 16) * 64) * 8;
  const size_t lds_mean = (size_t)std::max(1, kRT / d) * d * 8;
  DH_DEV_MEMO(attr_lds);
  DH_DEV_MEMO(attr_eig);
  DH_DEV_MEMO(attr_part);
  if (lds > attr_lds) {
    if (!hip_ok(ctx,
                hipFuncSetAttribute((const void*)wide_single_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "hipFuncSetAttribute(wide LDS)"))
      return DH_ERR_HIP;
    attr_lds = lds;
  }
  if (!split) {
    hipLaunchKernelGGL(wide_single_kernel, dim3(runs), dim3(kRT), lds, ctx->stream, a);
  } else {
    if (eig_lds > attr_eig) {
      if (!hip_ok(ctx,
                  hipFuncSetAttribute((const void*)wide_eig_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)eig_lds),
                  "hipFuncSetAttribute(wide eig LDS)") ||
          !hip_ok(ctx,
                  hipFuncSetAttribute((const void*)wide_eig2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)eig_lds),
                  "hipFuncSetAttribute(wide eig2 LDS)"))
        return DH_ERR_HIP;
      attr_eig = eig_lds;
    }
    if (std::max(lds_part, lds_mean) > attr_part) {
      const int want = (int)std::max(lds_part, lds_mean);
      if (!hip_ok(ctx, hipFuncSetAttribute((const void*)wide_cov_part_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, want), "attr(cov part)") ||
          !hip_ok(ctx, hipFuncSetAttribute((const void*)wide_fmax_part_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, want), "attr(fmax part)") ||
          !hip_ok(ctx, hipFuncSetAttribute((const void*)wide_mean_part_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, want), "attr(mean part)"))
        return DH_ERR_HIP;
      attr_part = (size_t)want;
    }
    if (!hip_ok(ctx, hipMemsetAsync(eig_int, 0, ints * runs, ctx->stream), "memset(eig counters)")) return DH_ERR_HIP;
    a.split_tail = 1;
    hipLaunchKernelGGL(wide_mean_part_kernel, dim3(runs * P), dim3(kRT), lds_mean, ctx->stream, a);
    hipLaunchKernelGGL(wide_cov_part_kernel, dim3(runs * P), dim3(kRT), lds_part, ctx->stream, a);
    hipLaunchKernelGGL(wide_cov_reduce_kernel, dim3((d * d + 255) / 256, runs), dim3(256), 0, ctx->stream, a);
    WideEigArgs g;
    g.cov = a.wscov;
    g.lam = lam_pre;
    g.V = a.wsV;
    g.xbuf = x_buf;
    g.bar = eig_int;
    g.rot = eig_int + runs;
    g.ok = eig_int + (size_t)runs * (1 + kEigMaxSweeps);
    g.D = d;
    g.B = B;
    g.b = b;
    g.dbg = a.dbg;
    g.active = active;
    g.V0t = g.Wt = nullptr;
    g.cold = nullptr;
    if (gram && warm_nells && !(getenv("DH_WIDE_WARM") && atoi(getenv("DH_WIDE_WARM")) == 0)) {
      hipLaunchKernelGGL(wide_warm_kernel, dim3(d, runs), dim3(256), 0, ctx->stream, d, a.wscov, axes, warm_nells, active,
                         ws_v0t, ws_wt, eig_cold);
      g.V0t = ws_v0t;
      g.Wt = ws_wt;
      g.cold = eig_cold;
    }
    if (!hip_ok(ctx, launch_all_resident(ctx, gram ? wide_eig2_kernel : wide_eig_kernel, dim3(runs * B), dim3(kRT), eig_lds, g),
                "wide_eig launch"))
      return DH_ERR_HIP;
    a.phase = 2;
    hipLaunchKernelGGL(wide_single_kernel, dim3(runs), dim3(kRT), lds, ctx->stream, a);
    hipLaunchKernelGGL(wide_fmax_part_kernel, dim3(runs * P), dim3(kRT), lds_part, ctx->stream, a);
    hipLaunchKernelGGL(wide_finish_kernel, dim3(runs), dim3(kRT), 0, ctx->stream, a);
  }
  return hip_ok(ctx, hipGetLastError(), "wide rebuild launch") ? DH_OK : DH_ERR_HIP;
}

int wide_single_launch(dh_ctx* ctx, int runs, const double* pts, int n, int d, int32_t* nells,
                       int32_t* status, double* ctrs, double* covs, double* ams, double* axes,
                       double* axlens, double* logvols) {
  const int rc0 = wide_single_enqueue(ctx, runs, pts, n, d, nullptr, status, ctrs, covs, ams, axes, axlens, logvols);
  if (rc0) return rc0;
  // nells = 1 per run (status decides validity)
  std::vector<int32_t> ones((size_t)runs, 1);
  if (!hip_ok(ctx, hipMemcpyAsync(nells, ones.data(), (size_t)runs * 4, hipMemcpyHostToDevice, ctx->stream),
              "H2D nells"))
    return DH_ERR_HIP;
  return hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync") ? DH_OK : DH_ERR_HIP;
}


// Ellipsoid.update of the runs with active[run] != 0, everything on the stream, nothing synchronised: the
// rebuild stage of the device-resident loop (ns.hip) above the register-resident dimensions.
__global__ void wide_mark_single_kernel(int runs, const int* __restrict__ active, int32_t* __restrict__ nells) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < runs && (!active || active[r])) nells[r] = 1;
}

int wide_single_launch_masked(dh_ctx* ctx, int runs, const double* pts, int n, int d, int32_t* nells,
                              int32_t* status, double* ctrs, double* covs, double* ams, double* axes,
                              double* axlens, double* logvols, const int* active) {
  const int rc0 = wide_single_enqueue(ctx, runs, pts, n, d, active, status, ctrs, covs, ams, axes, axlens, logvols, nells);
  if (rc0) return rc0;
  hipLaunchKernelGGL(wide_mark_single_kernel, dim3((runs + 63) / 64), dim3(64), 0, ctx->stream, runs, active, nells);
  return hip_ok(ctx, hipGetLastError(), "wide masked rebuild launch") ? DH_OK : DH_ERR_HIP;
}

// ---------------------------------------------------------------------------
// MultiEllipsoid.update at wide D (bounding.py:632-686, 1464-1563).  Nodes are big here (a split
// needs 2 x 2D points per side), so the tree has a handful of nodes: the recursion runs on the host
// exactly as the reference writes it, every node's work on the device -- the ellipsoid of a point
// subset is the multi-workgroup rebuild above on a gathered copy, k-means is one workgroup.

// points.std(axis=0) (ddof = 0): one thread per dimension, two passes in point order
__global__ void wide_std_kernel(const double* __restrict__ pts, int n, int D, double* __restrict__ scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D) return;
  double m = 0.0;
  for (int p = 0; p < n; ++p) m += pts[(size_t)p * D + j];
  m /= (double)n;
  double v = 0.0;
  for (int p = 0; p < n; ++p) {
    const double e = pts[(size_t)p * D + j] - m;
    v = fma(e, e, v);
  }
  scale[j] = sqrt(v / (double)n);
}

__global__ void wide_gather_kernel(const double* __restrict__ pts, const int32_t* __restrict__ idx, int cnt, int D,
                                   double* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)cnt * D) return;
  const int p = (int)(e / D), j = (int)(e - (size_t)p * D);
  out[e] = pts[(size_t)idx[p] * D + j];
}

// scipy.cluster.vq.kmeans2(points / scale, k = seeds / scale, iter = 10, minit = 'matrix') with the
// major-axis endpoints of the node's ellipsoid as seeds (bounding.py:1500-1514): vq = nearest
// centroid (distance summed in dimension order, strict '<': the lower index wins ties), then
// update_cluster_means = per-cluster sums IN POINT ORDER / counts, an empty cluster keeping its
// centroid.  The labels returned are those of the last vq.  One workgroup.
__global__ void __launch_bounds__(kRT) wide_kmeans_kernel(const double* __restrict__ pts, int cnt, int D,
                                                         const double* __restrict__ scale,
                                                         const double* __restrict__ ctr,
                                                         const double* __restrict__ axes,
                                                         const double* __restrict__ axlens, int32_t* labels,
                                                         int32_t* n0_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* cen = (double*)smem;  // 2 x D
  double* isc = cen + 2 * D;    // D : the scale
  __shared__ int s_best, s_cnt[kRT / 64];
  const int t = threadIdx.x;
  if (t == 0) {
    int best = 0;
    double bl = axlens[0];
    for (int k = 1; k < D; ++k)
      if (axlens[k] > bl) {
        bl = axlens[k];
        best = k;
      }
    s_best = best;
  }
  for (int j = t; j < D; j += kRT) isc[j] = scale[j];
  __syncthreads();
  for (int j = t; j < D; j += kRT) {
    const double v = axes[(size_t)j * D + s_best];
    cen[j] = (ctr[j] - v) / isc[j];
    cen[D + j] = (ctr[j] + v) / isc[j];
  }
  __syncthreads();
  int n0 = 0;
  for (int it = 0; it < 10; ++it) {
    int mine0 = 0;
    for (int p = t; p < cnt; p += kRT) {
      const double* x = pts + (size_t)p * D;
      double d0 = 0.0, d1 = 0.0;
      for (int j = 0; j < D; ++j) {
        const double xv = x[j] / isc[j];
        const double e0 = xv - cen[j], e1 = xv - cen[D + j];
        d0 = fma(e0, e0, d0);
        d1 = fma(e1, e1, d1);
      }
      const int lb = d1 < d0 ? 1 : 0;
      labels[p] = lb;
      mine0 += lb == 0;
    }
    // count of label 0
    for (int o = 32; o > 0; o >>= 1) mine0 += __shfl_xor(mine0, o);
    if ((t & 63) == 0) s_cnt[t >> 6] = mine0;
    __threadfence_block();
    __syncthreads();
    n0 = 0;
    for (int w = 0; w < kRT / 64; ++w) n0 += s_cnt[w];
    const int n1 = cnt - n0;
    // cluster sums in point order: thread (c, j)
    double sum = 0.0;
    const int c = t >= D ? 1 : 0, j = t - c * D;
    if (t < 2 * D) {
      const double sc = isc[j];
      for (int p0 = 0; p0 < cnt; p0 += 8) {
        int lb[8];
        double xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int p = p0 + q;
          lb[q] = p < cnt ? labels[p] : -1;
          xv[q] = p < cnt ? pts[(size_t)p * D + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (lb[q] == c) sum += xv[q] / sc;
      }
    }
    __syncthreads();
    if (t < 2 * D) {
      const int nc = c == 0 ? n0 : n1;
      if (nc > 0) cen[c * D + j] = sum / (double)nc;
    }
    __syncthreads();
  }
  if (t == 0) *n0_out = n0;
}

namespace {

struct WideSlots {  // device arrays of node ellipsoids
  double *ctrs, *covs, *ams, *axes, *axlens, *logvols;
  int32_t* status;
  int cap, used;
};

struct WideTree {
  dh_ctx* ctx;
  const double* pts;  // all points of the run (device)
  int n, d;
  const double* scale;  // device, d
  WideSlots sl;
  int32_t* d_labels;  // n
  int32_t* d_n0;
  int32_t* d_idx;     // n
  std::vector<double> logvol;  // host copy per slot
  int nnodes = 0;
  int err = DH_OK;
};

// bounding ellipsoid of `cnt` contiguous points -> a fresh slot; returns the slot or -1
int wide_node_ell(WideTree& T, const double* p, int cnt) {
  WideSlots& S = T.sl;
  if (S.used >= S.cap) {
    T.err = DH_ERR_NOMEM;
    return -1;
  }
  const int s = S.used++;
  const size_t d = T.d, dd = d * d;
  int rc = wide_single_enqueue(T.ctx, 1, p, cnt, T.d, nullptr, S.status + s, S.ctrs + s * d, S.covs + s * dd, S.ams + s * dd,
                               S.axes + s * dd, S.axlens + s * d, S.logvols + s);
  if (rc) {
    T.err = rc;
    return -1;
  }
  int32_t st = DH_OK;
  double lv = 0.0;
  if (!hip_ok(T.ctx, hipMemcpyAsync(&st, S.status + s, 4, hipMemcpyDeviceToHost, T.ctx->stream), "D2H status") ||
      !hip_ok(T.ctx, hipMemcpyAsync(&lv, S.logvols + s, 8, hipMemcpyDeviceToHost, T.ctx->stream), "D2H logvol") ||
      !hip_ok(T.ctx, hipStreamSynchronize(T.ctx->stream), "sync")) {
    T.err = DH_ERR_HIP;
    return -1;
  }
  if (st != DH_OK) {
    T.err = st;
    return -1;
  }
  if ((int)T.logvol.size() <= s) T.logvol.resize(s + 1);
  T.logvol[s] = lv;
  return s;
}

double host_logaddexp(double x, double y) {
  if (x == y) return x + 0.6931471805599453;
  const double dlt = x - y;
  if (dlt > 0) return x + log1p(exp(-dlt));
  if (dlt <= 0) return y + log1p(exp(dlt));
  return x + y;
}

// _bounding_ellipsoids (bounding.py:1464-1563): node = global indices `idx`, its points gathered in
// `p` (device), its ellipsoid in slot `s`.  Appends the leaves (slot, indices) in list order.
void wide_split(WideTree& T, const std::vector<int32_t>& idx, const double* p, int s,
                std::vector<std::pair<int, std::vector<int32_t>>>& out) {
  ++T.nnodes;
  const int cnt = (int)idx.size(), d = T.d, min_size = 2 * d;
  if (T.err || cnt < 2 * min_size) {
    out.emplace_back(s, idx);
    return;
  }
  const size_t dd = (size_t)d * d;
  hipLaunchKernelGGL(wide_kmeans_kernel, dim3(1), dim3(kRT), (size_t)3 * d * 8, T.ctx->stream, p, cnt, d, T.scale,
                     T.sl.ctrs + (size_t)s * d, T.sl.axes + (size_t)s * dd, T.sl.axlens + (size_t)s * d, T.d_labels,
                     T.d_n0);
  std::vector<int32_t> lab((size_t)cnt);
  if (!hip_ok(T.ctx, hipGetLastError(), "wide kmeans launch") ||
      !hip_ok(T.ctx, hipMemcpyAsync(lab.data(), T.d_labels, (size_t)cnt * 4, hipMemcpyDeviceToHost, T.ctx->stream),
              "D2H labels") ||
      !hip_ok(T.ctx, hipStreamSynchronize(T.ctx->stream), "sync")) {
    T.err = DH_ERR_HIP;
    return;
  }
  std::vector<int32_t> kid[2];
  for (int i = 0; i < cnt; ++i) kid[lab[i] ? 1 : 0].push_back(idx[i]);
  if ((int)std::min(kid[0].size(), kid[1].size()) < min_size) {
    out.emplace_back(s, idx);
    return;
  }
  double* buf[2] = {nullptr, nullptr};
  int ks[2] = {-1, -1};
  for (int c = 0; c < 2 && !T.err; ++c) {
    const size_t m = kid[c].size();
    if (!hip_ok(T.ctx, hipMalloc((void**)&buf[c], m * d * 8), "hipMalloc(child points)")) {
      T.err = DH_ERR_NOMEM;
      break;
    }
    if (!hip_ok(T.ctx, hipMemcpyAsync(T.d_idx, kid[c].data(), m * 4, hipMemcpyHostToDevice, T.ctx->stream),
                "H2D child indices")) {
      T.err = DH_ERR_HIP;
      break;
    }
    const size_t tot = m * d;
    hipLaunchKernelGGL(wide_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, T.ctx->stream, T.pts,
                       T.d_idx, (int)m, d, buf[c]);
    // (the index staging buffer is reused by the sibling: finish this gather first)
    if (!hip_ok(T.ctx, hipStreamSynchronize(T.ctx->stream), "sync")) {
      T.err = DH_ERR_HIP;
      break;
    }
    ks[c] = wide_node_ell(T, buf[c], (int)m);
  }
  if (!T.err) {
    const int nparam = (d * (d + 3)) / 2;
    const double dec = nparam * log((double)cnt) / (double)cnt;
    std::vector<std::pair<int, std::vector<int32_t>>> sub;
    wide_split(T, kid[0], buf[0], ks[0], sub);
    wide_split(T, kid[1], buf[1], ks[1], sub);
    if (!T.err) {
      bool accept = (host_logaddexp(T.logvol[ks[0]], T.logvol[ks[1]]) - T.logvol[s]) < -dec;
      if (!accept) {
        // scipy.special.logsumexp over the leaves
        double mx = -INFINITY;
        for (auto& e : sub) mx = std::max(mx, T.logvol[e.first]);
        double acc = 0.0;
        for (auto& e : sub) acc += exp(T.logvol[e.first] - mx);
        accept = (log(acc) + mx - T.logvol[s]) < -dec * ((double)sub.size() - 1.0);
      }
      if (accept)
        for (auto& e : sub) out.push_back(std::move(e));
      else
        out.emplace_back(s, idx);
    }
  }
  for (int c = 0; c < 2; ++c)
    if (buf[c]) (void)hipFree(buf[c]);
}

}  // namespace

int wide_multi_launch(dh_ctx* ctx, int runs, const double* pts, int n, int d, int max_ells, int32_t* nells,
                      int32_t* status, double* ctrs, double* covs, double* ams, double* axes, double* axlens,
                      double* logvols, int32_t* leaf_of_point, int32_t* nnodes, const int* active) {
  if (d > kWideMaxD) return fail(ctx, DH_ERR_ARG, "rebuild: d=%d exceeds the wide-D limit %d", d, kWideMaxD);
  const size_t dd = (size_t)d * d;
  // run mask of the device-resident loop (device array): the recursion is driven from the host, so the mask comes here
  std::vector<int> h_active;
  if (active) {
    h_active.resize((size_t)runs);
    if (!hip_ok(ctx, hipMemcpyAsync(h_active.data(), active, (size_t)runs * 4, hipMemcpyDeviceToHost, ctx->stream), "D2H run mask") ||
        !hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync"))
      return DH_ERR_HIP;
    bool any = false;
    for (int v : h_active) any = any || v != 0;
    if (!any) return DH_OK;
  }
  const int cap = std::max(3, n / d + 3);  // every split makes two children of >= 2d points
  char* pool = nullptr;
  const size_t per = (3 * dd + 2 * (size_t)d + 1) * 8 + 4;
  const size_t bytes = per * cap + (size_t)d * 8 + (size_t)n * 8 + 64 + 4096;
  if (!hip_ok(ctx, hipMalloc((void**)&pool, bytes), "hipMalloc(wide multi scratch)")) return DH_ERR_NOMEM;
  int rc = DH_OK;
  for (int run = 0; run < runs && rc == DH_OK; ++run) {
    if (active && !h_active[(size_t)run]) continue;
    WideTree T;
    T.ctx = ctx;
    T.pts = pts + (size_t)run * n * d;
    T.n = n;
    T.d = d;
    double* w = (double*)pool;
    T.sl.ctrs = w;
    w += (size_t)cap * d;
    T.sl.covs = w;
    w += (size_t)cap * dd;
    T.sl.ams = w;
    w += (size_t)cap * dd;
    T.sl.axes = w;
    w += (size_t)cap * dd;
    T.sl.axlens = w;
    w += (size_t)cap * d;
    T.sl.logvols = w;
    w += cap;
    double* d_scale = w;
    w += d;
    T.scale = d_scale;
    T.sl.status = (int32_t*)w;
    T.d_labels = T.sl.status + cap;
    T.d_idx = T.d_labels + n;
    T.d_n0 = T.d_idx + n;
    T.sl.cap = cap;
    T.sl.used = 0;
    int32_t st = DH_OK, m = 0;
    std::vector<std::pair<int, std::vector<int32_t>>> leaves;
    if (n <= 1) {
      st = DH_ERR_VALUE;
    } else {
      const int root = wide_node_ell(T, T.pts, n);
      if (root >= 0) {
        hipLaunchKernelGGL(wide_std_kernel, dim3((d + 63) / 64), dim3(64), 0, ctx->stream, T.pts, n, d, d_scale);
        std::vector<int32_t> all((size_t)n);
        for (int i = 0; i < n; ++i) all[i] = i;
        wide_split(T, all, T.pts, root, leaves);
      }
      st = T.err;
      if (st == DH_ERR_HIP || st == DH_ERR_ARG) {
        rc = st;
        break;
      }
    }
    if (st == DH_OK && (int)leaves.size() > max_ells) st = DH_ERR_NOMEM;
    if (st == DH_OK) {
      m = (int)leaves.size();
      std::vector<int32_t> lop((size_t)n, 0);
      for (int e = 0; e < m; ++e) {
        const size_t s = (size_t)leaves[e].first, o = (size_t)run * max_ells + e;
        bool ok = hip_ok(ctx, hipMemcpyAsync(ctrs + o * d, T.sl.ctrs + s * d, (size_t)d * 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D") &&
                  hip_ok(ctx, hipMemcpyAsync(covs + o * dd, T.sl.covs + s * dd, dd * 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D") &&
                  hip_ok(ctx, hipMemcpyAsync(ams + o * dd, T.sl.ams + s * dd, dd * 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D") &&
                  hip_ok(ctx, hipMemcpyAsync(axes + o * dd, T.sl.axes + s * dd, dd * 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D") &&
                  hip_ok(ctx, hipMemcpyAsync(axlens + o * d, T.sl.axlens + s * d, (size_t)d * 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D") &&
                  hip_ok(ctx, hipMemcpyAsync(logvols + o, T.sl.logvols + s, 8, hipMemcpyDeviceToDevice, ctx->stream), "D2D");
        if (!ok) {
          rc = DH_ERR_HIP;
          break;
        }
        for (int32_t i : leaves[e].second) lop[(size_t)i] = e;
      }
      if (rc) break;
      // every point inside some ellipsoid (bounding.py:683-685)
      int32_t* d_count = T.d_labels;
      rc = wide_contains_launch(ctx, T.pts, n, d, ctrs + (size_t)run * max_ells * d, ams + (size_t)run * max_ells * dd, m,
                                0, d_count, nullptr, nullptr);
      if (rc) break;
      std::vector<int32_t> cntv((size_t)n);
      if (!hip_ok(ctx, hipMemcpyAsync(cntv.data(), d_count, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream), "D2H") ||
          !hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) {
        rc = DH_ERR_HIP;
        break;
      }
      for (int i = 0; i < n; ++i)
        if (cntv[i] < 1) st = DH_ERR_REGION;
      if (leaf_of_point &&
          !hip_ok(ctx, hipMemcpyAsync(leaf_of_point + (size_t)run * n, lop.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream), "H2D")) {
        rc = DH_ERR_HIP;
        break;
      }
    }
    const int32_t nn = T.nnodes;
    if (!hip_ok(ctx, hipMemcpyAsync(nells + run, &m, 4, hipMemcpyHostToDevice, ctx->stream), "H2D") ||
        !hip_ok(ctx, hipMemcpyAsync(status + run, &st, 4, hipMemcpyHostToDevice, ctx->stream), "H2D") ||
        (nnodes && !hip_ok(ctx, hipMemcpyAsync(nnodes + run, &nn, 4, hipMemcpyHostToDevice, ctx->stream), "H2D")) ||
        !hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) {
      rc = DH_ERR_HIP;
      break;
    }
  }
  (void)hipFree(pool);
  return rc;
}

}  // namespace dh
