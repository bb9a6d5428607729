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
  const double* like_par;   // [c, ...]
  const double* prior_par;  // [a, b] / [mu, sigma]
};

// v = ndtri(u): scipy.special.ndtri (Cephes) is the host function.  On the
// device: ocml's erfcinv followed by one Newton step on erfc, which lands
// within a few ulp of the correctly rounded value over (0,1).
__device__ __noinline__ double ndtri_dev(double p) {
  double x = -1.4142135623730951 * erfcinv(2.0 * p);
  double f = 0.5 * erfc(-x * 0.7071067811865476) - p;
  double pdf = 0.3989422804014327 * exp(-0.5 * x * x);
  if (pdf > 1e-300) x -= f / pdf;
  return x;
}

// n = live dimension count (<= N); entries i >= n of v are set to 0.
// `tmp` is per-lane LDS scratch ([dim][64 lanes]) used to keep the
// transcendental prior out of the unrolled register code.
template <int N, bool FULL>
__device__ __forceinline__ void prior_transform(const ProblemDev& P, const double (&u)[N],
                                                double (&v)[N], int n, double* tmp) {
  if (P.prior_id == PRIOR_AFFINE) {
    cdptr pp = as_const(P.prior_par);
    const double a = pp[0], b = pp[1];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (FULL || i < n) ? a * (2.0 * u[i] - 1.0) + b : 0.0;
  } else if (P.prior_id == PRIOR_NORMAL) {
    cdptr pp = as_const(P.prior_par);
    const double mu = pp[0], sg = pp[1];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i * 64 + lane] = u[i];
#pragma unroll 1
    for (int i = 0; i < n; ++i) tmp[i * 64 + lane] = mu + sg * ndtri_dev(tmp[i * 64 + lane]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (FULL || i < n) ? tmp[i * 64 + lane] : 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (FULL || i < n) ? u[i] : 0.0;
  }
}

template <int N, bool FULL>
__device__ __forceinline__ double loglike(const ProblemDev& P, const double (&v)[N], int n,
                                          double* tmp) {
  cdptr lp = as_const(P.like_par);
  if (P.like_id == LIKE_GAUSS_PREC) {
    // -0.5 v^T P v + c, P symmetric: diagonal + 2 * strict upper triangle
    cdptr A = lp + 1;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (FULL || i < n) {
        double r = 0.5 * A[i * n + i] * v[i];
#pragma unroll
        for (int j = i + 1; j < N; ++j)
          if (FULL || j < n) r = fma(A[i * n + j], v[j], r);
        q = fma(v[i], r, q);
      }
    }
    return lp[0] - q;
  } else if (P.like_id == LIKE_EGGBOX) {
    const double tmax = lp[0];
    double prod = 1.0;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i * 64 + lane] = v[i];
#pragma unroll 1
    for (int i = 0; i < n; ++i) prod *= cos((2.0 * tmax * tmp[i * 64 + lane] - tmax) / 2.0);
    double b = 2.0 + prod;
    double b2 = b * b;
    return b2 * b2 * b;
  } else {
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) q = fma(v[i], v[i], q);  // padded entries are 0
    return lp[0] - 0.5 * q;
  }
}

}  // namespace dh
