// In-kernel prior transforms and log-likelihoods of the BASELINE problems.
//
// dynesty evaluates prior_transform(u) and loglikelihood(v) once per proposal
// inside the proposal loop (reference internal_samplers.py:957-958, 1116-1117,
// 328-329).  For the device path the two user callbacks are replaced by the
// ids below; the host twin (same operation order) is dynesty_amd/problems.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dh {

enum : int { LIKE_GAUSS_IID = 0, LIKE_GAUSS_PREC = 1, LIKE_EGGBOX = 2 };
enum : int { PRIOR_IDENTITY = 0, PRIOR_AFFINE = 1, PRIOR_NORMAL = 2 };

// Kernels are specialised on the (likelihood, prior) pair so the proposal loop
// carries no dispatch; KIND_GENERIC keeps a runtime switch for other pairs.
enum : int {
  KIND_GENERIC = 0,
  KIND_PREC_AFFINE = 1,     // C2: correlated Normal, uniform box prior
  KIND_IID_AFFINE = 2,      // C1: iid Normal, uniform box prior
  KIND_EGGBOX_IDENTITY = 3, // C3
  KIND_IID_NORMAL = 4,      // C4: iid Normal, Normal prior (ndtri)
  KIND_COUNT = 5
};

__host__ __device__ inline int problem_kind(int like_id, int prior_id) {
  if (like_id == LIKE_GAUSS_PREC && prior_id == PRIOR_AFFINE) return KIND_PREC_AFFINE;
  if (like_id == LIKE_GAUSS_IID && prior_id == PRIOR_AFFINE) return KIND_IID_AFFINE;
  if (like_id == LIKE_EGGBOX && prior_id == PRIOR_IDENTITY) return KIND_EGGBOX_IDENTITY;
  if (like_id == LIKE_GAUSS_IID && prior_id == PRIOR_NORMAL) return KIND_IID_NORMAL;
  return KIND_GENERIC;
}

// Read-only, wave-uniform operands (proposal frames, precision matrices) are
// addressed through the constant address space so the backend always selects
// scalar-cache loads (s_load_dwordxN) and feeds v_fma_f64 from SGPRs.
typedef const __attribute__((address_space(4))) double* cdptr;
__device__ __forceinline__ cdptr as_const(const double* p) {
  return (cdptr)(unsigned long long)p;
}

// Passed by value to kernels; the parameter blocks live in device memory.
struct ProblemDev {
  int like_id;
  int prior_id;
  int ndim;
  const double* like_par;   // [c, P row-major (n x n)] / [c] / [tmax]
  const double* prior_par;  // [a, b] / [mu, sigma]
  const double* prec_t;     // GAUSS_PREC: P transposed + zero padded to the kernel's N
};

template <int KIND>
__device__ __forceinline__ int like_of(const ProblemDev& P) {
  return KIND == KIND_GENERIC         ? P.like_id
         : KIND == KIND_PREC_AFFINE   ? LIKE_GAUSS_PREC
         : KIND == KIND_EGGBOX_IDENTITY ? LIKE_EGGBOX
                                        : LIKE_GAUSS_IID;
}
template <int KIND>
__device__ __forceinline__ int prior_of(const ProblemDev& P) {
  return KIND == KIND_GENERIC           ? P.prior_id
         : KIND == KIND_EGGBOX_IDENTITY ? PRIOR_IDENTITY
         : KIND == KIND_IID_NORMAL      ? PRIOR_NORMAL
                                        : PRIOR_AFFINE;
}

// v = ndtri(u): scipy.special.ndtri (Cephes) is the host function.  On the device ocml's
// erfcinv alone is within 8.3e-16 relative of it over (1e-300, 1 - 1e-15) (measured:
// tools/micro/ndtri_acc.hip); a Newton step on erfc, tried first, only adds cancellation error
// near p = 1/2 (1.5e-11 relative) and doubles the cost.
__device__ __forceinline__ double ndtri_dev(double p) { return -1.4142135623730951 * erfcinv(2.0 * p); }

// acc[i] += sum_j MT[j*N + i] * xs[j*64 + lane]   for i in [I0, I0+NI), j < nj.
// MT is wave-uniform (scalar loads); the row for j+1 is requested before the
// FMAs of row j are issued so the scalar-cache latency overlaps the math.
template <int N, int I0, int NI>
__device__ __forceinline__ void matvec_block(cdptr MT, const double* xs, int lane, int nj,
                                             double (&acc)[N]) {
  double cur[NI], nxt[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) cur[i] = MT[I0 + i];
  double d = xs[lane];
#pragma unroll 1
  for (int j = 0; j < nj; ++j) {
    const int jn = (j + 1 < nj) ? j + 1 : j;
#pragma unroll
    for (int i = 0; i < NI; ++i) nxt[i] = MT[jn * N + I0 + i];
    const double dn = xs[jn * 64 + lane];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[I0 + i] = fma(cur[i], d, acc[I0 + i]);
#pragma unroll
    for (int i = 0; i < NI; ++i) cur[i] = nxt[i];
    d = dn;
  }
}

// acc += MT^T-sweep over all N outputs, split so a double-buffered row block
// fits the SGPR file (2 x 13 doubles = 52 SGPRs).
template <int N>
__device__ __forceinline__ void matvec_sgpr(cdptr MT, const double* xs, int lane, int nj,
                                            double (&acc)[N]) {
#ifndef DH_MV_ONE
#define DH_MV_ONE 25
#endif
  if constexpr (N <= DH_MV_ONE) {
    matvec_block<N, 0, N>(MT, xs, lane, nj, acc);
  } else if constexpr (N <= 26) {
    constexpr int H = (N + 1) / 2;
    matvec_block<N, 0, H>(MT, xs, lane, nj, acc);
    matvec_block<N, H, N - H>(MT, xs, lane, nj, acc);
  } else {
    constexpr int T = (N + 2) / 3;
    matvec_block<N, 0, T>(MT, xs, lane, nj, acc);
    matvec_block<N, T, T>(MT, xs, lane, nj, acc);
    matvec_block<N, 2 * T, N - 2 * T>(MT, xs, lane, nj, acc);
  }
}

// v = prior_transform(u), written to the per-lane LDS column `sv`
// ([dim][64 lanes]); entries i >= n are 0.  The rolled loop keeps the
// transcendental priors out of the unrolled register code.
template <int N, bool FULL, int KIND>
__device__ __forceinline__ void prior_to_lds(const ProblemDev& P, const double (&u)[N], int n,
                                             double* sv, int lane) {
  const int pid = prior_of<KIND>(P);
  if (pid == PRIOR_AFFINE) {
    cdptr pp = as_const(P.prior_par);
    const double a = pp[0], b = pp[1];
#pragma unroll
    for (int i = 0; i < N; ++i) sv[i * 64 + lane] = (FULL || i < n) ? a * (2.0 * u[i] - 1.0) + b : 0.0;
  } else if (pid == PRIOR_NORMAL) {
    cdptr pp = as_const(P.prior_par);
    const double mu = pp[0], sg = pp[1];
#pragma unroll
    for (int i = 0; i < N; ++i) sv[i * 64 + lane] = u[i];
#pragma unroll 1
    for (int i = 0; i < N; ++i)
      sv[i * 64 + lane] = (FULL || i < n) ? mu + sg * ndtri_dev(sv[i * 64 + lane]) : 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) sv[i * 64 + lane] = (FULL || i < n) ? u[i] : 0.0;
  }
}

// log-likelihood of the v staged in `sv` by prior_to_lds.  `w` is register
// scratch for the precision mat-vec.
template <int N, bool FULL, int KIND>
__device__ __forceinline__ double loglike_lds(const ProblemDev& P, int n, const double* sv, int lane,
                                              double (&w)[N]) {
  cdptr lp = as_const(P.like_par);
  const int lid = like_of<KIND>(P);
  if (lid == LIKE_GAUSS_PREC) {
    // -0.5 v^T P v + c, P symmetric: q = sum_i v_i (P_ii v_i / 2 + sum_{j>i} P_ij v_j)
    // (half the FMAs of a full mat-vec); rows of P come through the scalar cache.
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = sv[i * 64 + lane];
    if constexpr (FULL) {
      cdptr A = lp + 1;
      double q0 = 0.0, q1 = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double r = 0.5 * A[i * N + i] * w[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) r = fma(A[i * N + j], w[j], r);
        if (i & 1)
          q1 = fma(w[i], r, q1);
        else
          q0 = fma(w[i], r, q0);
      }
      return lp[0] - (q0 + q1);
    } else {
      cdptr A = as_const(P.prec_t);  // padded to N x N with zeros
      double q = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double r = 0.5 * A[i * N + i] * w[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j) r = fma(A[i * N + j], w[j], r);
        q = fma(w[i], r, q);
      }
      return lp[0] - q;
    }
  } else if (lid == LIKE_EGGBOX) {
    const double tmax = lp[0];
    double prod = 1.0;
#pragma unroll 1
    for (int i = 0; i < n; ++i) prod *= cos((2.0 * tmax * sv[i * 64 + lane] - tmax) / 2.0);
    const double b = 2.0 + prod;
    const double b2 = b * b;
    return b2 * b2 * b;
  } else {
    double q0 = 0.0, q1 = 0.0;
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {  // padded entries are 0
      const double a = sv[i * 64 + lane], b = sv[(i + 1) * 64 + lane];
      q0 = fma(a, a, q0);
      q1 = fma(b, b, q1);
    }
    if (N & 1) {
      const double a = sv[(N - 1) * 64 + lane];
      q0 = fma(a, a, q0);
    }
    return lp[0] - 0.5 * (q0 + q1);
  }
}

}  // namespace dh
