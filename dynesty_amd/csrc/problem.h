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

// periodic wrap and reflection of one unit-cube coordinate (utils.py:1053-1078 apply_reflect; np.mod(x, 1))
__device__ __forceinline__ double wrap01(double x) { return x - floor(x); }
__device__ __forceinline__ double reflect01(double x) {
  const double m2 = x - 2.0 * floor(x * 0.5);  // np.mod(x, 2)
  const double m1 = wrap01(x);
  return (m2 < 1.0) ? m1 : 1.0 - m1;
}

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
// A wave-uniform pointer the compiler does not KNOW to be uniform -- e.g. one indexed by the
// readfirstlane value of a waterfall loop: inside `if (cur == my_frame)` LLVM's equality propagation
// rewrites `cur` (SGPR) to `my_frame` (VGPR) and the matrix rows then arrive as 13 per-lane
// global_load_dwordx4 per row instead of scalar loads (found in the ISA of rwalk_kernel; the frame product
// was 2x slower than its SGPR form).  Reading the two halves back through readfirstlane pins it to SGPRs.
__device__ __forceinline__ cdptr as_const_uniform(const double* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (cdptr)(((unsigned long long)hi << 32) | lo);
}

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

// The same function for kernels where it is the critical path (one wavefront per walker at wide D:
// a lane owns a few coordinates and ocml's erfcinv is a long dependent chain behind divergent
// branches).  Wichura's AS 241 (PPND16, Appl. Statist. 37 (1988) 477): rational approximations in
// r = 0.180625 - q^2 for |p - 1/2| <= 0.425 and in sqrt(-ln min(p, 1-p)) - 1.6 beyond; both are
// evaluated branch-free (straight-line code: N independent chains for the scheduler to
// interleave) and selected.  Measured against scipy.special.ndtri: <= 1.1e-15 relative
// (tools/ndtri_as241_check.py).  p < exp(-25) (or p outside (0, 1)) takes ndtri_dev.
// ln x for x > 0 in plain fp64 (fdlibm's e_log.c scheme: x = 2^k (1 + f), s = f / (2 + f), a degree-7
// polynomial in s^2, k ln2 in two parts): <= 1 ulp against NumPy's log over (e^-30, 1)
// (tools/ndtri_as241_check.py), a third of the instructions of ocml's double-double log -- the
// logarithm was the largest part of the tail branch below.
__device__ __forceinline__ double log_pos(double x) {
  int e;
  double m = frexp(x, &e);  // [0.5, 1)
  const bool lo = m < 0.7071067811865476;
  m = lo ? m + m : m;
  const double k = (double)(lo ? e - 1 : e);
  const double f = m - 1.0;
  const double s = f / (2.0 + f), z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return k * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + k * 1.90821492927058770002e-10)) - f);
}

__device__ __forceinline__ double ndtri_as241_core(double p, bool* far) {
  const double q = p - 0.5;
  double r = 0.180625 - q * q;
  const double cn =
      (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
           45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
        133.14166789178437745) * r + 3.387132872796366608);
  const double cd =
      (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r +
           21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
        42.313330701600911252) * r + 1.0);
  const double pm = q < 0.0 ? p : 1.0 - p;
  const double rt = sqrt(-log_pos(pm));
  *far = !(rt <= 5.0) || !(pm > 0.0);
  r = rt - 1.6;
  const double tn =
      (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r +
           1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
        4.6303378461565452959) * r + 1.42343711074968357734);
  const double td =
      (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r +
           .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r +
        2.05319162663775882187) * r + 1.0);
  const bool central = fabs(q) <= 0.425;
  const double num = central ? q * cn : (q < 0.0 ? -tn : tn);
  const double den = central ? cd : td;
  return num / den;
}

// (the far tail -- p < e^-25, or outside (0, 1) -- is a call: inlined, ocml's erfcinv sets the register need of every
// kernel that merely might meet such a coordinate)
__device__ __attribute__((noinline)) double ndtri_far(double p) { return ndtri_dev(p); }

// The two halves of AS 241 on their own, for callers that route the coordinates themselves (wide.hip): the central
// rational approximation (|p - 1/2| <= 0.425: 85 % of the coordinates of a draw from the prior) is a third of the
// work of the tail's (logarithm, square root, a second pair of polynomials).  Same expressions as
// ndtri_as241_core, term for term: a coordinate gets the same bits by either route.
__device__ __forceinline__ double ndtri_as241_central(double p) {
  const double q = p - 0.5;
  const double r = 0.180625 - q * q;
  const double cn =
      (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
           45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
        133.14166789178437745) * r + 3.387132872796366608);
  const double cd =
      (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r +
           21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
        42.313330701600911252) * r + 1.0);
  return q * cn / cd;
}
__device__ __forceinline__ double ndtri_as241_tail(double p, bool* far) {
  const double q = p - 0.5;
  const double pm = q < 0.0 ? p : 1.0 - p;
  const double rt = sqrt(-log_pos(pm));
  *far = !(rt <= 5.0) || !(pm > 0.0);
  const double r = rt - 1.6;
  const double tn =
      (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r +
           1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
        4.6303378461565452959) * r + 1.42343711074968357734);
  const double td =
      (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r +
           .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r +
        2.05319162663775882187) * r + 1.0);
  return (q < 0.0 ? -tn : tn) / td;
}

template <int N>
__device__ __forceinline__ void ndtri_n(const double (&p)[N], double (&out)[N]) {
  bool far[N], any_far = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    out[i] = ndtri_as241_core(p[i], &far[i]);
    any_far = any_far || far[i];
  }
  if (__builtin_expect(any_far, 0)) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (far[i]) out[i] = ndtri_far(p[i]);
  }
}

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
      sv[i * 64 + lane] = (FULL || i < n) ? mu + sg * ndtri_far(sv[i * 64 + lane]) : 0.0;  // (a call: see ndtri_far)
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) sv[i * 64 + lane] = (FULL || i < n) ? u[i] : 0.0;
  }
}

// log-likelihood of the v staged in `sv` by prior_to_lds.  `w` is register
// scratch for the precision mat-vec.
// SB > 0: a scheduling barrier after every SB rows of the triangular form (LIKE_GAUSS_PREC, FULL).  Without
// it the scheduler clusters the scalar loads of all 325 unrolled FMAs; in the PCG64 rwalk kernel, whose
// registers are exhausted, that meant ~690 SGPR spills as soon as the frame rows also came through the
// scalar cache.  With SB = 4 that kernel runs 1.44 -> 1.24 ms; the Philox kernel is faster without (0.86 vs
// 1.1-1.2 ms), so the caller chooses.
template <int N, bool FULL, int KIND, int SB = 0>
__device__ __forceinline__ double loglike_lds(const ProblemDev& P, int n, const double* sv, int lane,
                                              double (&w)[N]) {
  cdptr lp = as_const(P.like_par);
  const int lid = like_of<KIND>(P);
  if (lid == LIKE_GAUSS_PREC) {
    // -0.5 v^T P v + c, P symmetric: q = sum_i v_i (P_ii v_i / 2 + sum_{j>i} P_ij v_j)
    // (half the FMAs of a full mat-vec); rows of P come through the scalar cache.
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = sv[i * 64 + lane];
#ifdef DH_LIKE_MATVEC
    if constexpr (FULL) {
      // y = P v with the rolled, double-buffered row loop of the frame product (P is symmetric, so its
      // rows are its columns); q = v . y / 2.  Twice the FMAs of the triangular form below, but that form
      // is 325 unrolled FMAs in one scheduling region whose scalar loads the compiler clusters and spills
      // (594 v_writelane + 598 v_readlane per step in the PCG64 kernel).
      double y[N];
#pragma unroll
      for (int i = 0; i < N; ++i) y[i] = 0.0;
      matvec_sgpr<N>(lp + 1, sv, lane, N, y);
      double q0 = 0.0, q1 = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (i & 1)
          q1 = fma(w[i], y[i], q1);
        else
          q0 = fma(w[i], y[i], q0);
      }
      return lp[0] - 0.5 * (q0 + q1);
    } else
#endif
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
        if constexpr (SB > 0) {
          if ((i % SB) == SB - 1) __builtin_amdgcn_sched_barrier(0);
        }
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
