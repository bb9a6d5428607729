// Wave-level symmetric eigensolver shared by rebuild.hip and friends.hip (device code; static
// functions, one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace dh_eig {

static __device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// ---- symmetric eigen-decomposition by one wavefront -------------------------
// A (D x LD, symmetric, destroyed: diagonal -> eigenvalues), V -> eigenvectors
// in columns.  Parallel-order cyclic Jacobi: every round rotates ceil(D/2)
// disjoint (p,q) pairs at once (round-robin tournament), so a sweep is D-1 (D)
// rounds of three LDS passes.  Called by wave 0 only; caller barriers after.
// Returns false if the matrix contains non-finite entries.
static __device__ bool jacobi_wave(double* A, double* V, int D, int LD, double* rc, double* rs, int* rp) {
  const int lane = threadIdx.x & 63;
  // power-of-two lane maps (no integer division in the hot loops):
  //   (row i, column j): j = lane & (JW-1), i strides by 64/JW
  //   (row i, pair k):   k = lane & (KW-1), i strides by 64/KW
  const int m = (D + 1) / 2;  // pairs per round
  int JW = 1;
  while (JW < D) JW <<= 1;  // <= 64
  int KW = 1;
  while (KW < m) KW <<= 1;  // <= 32
  const int jj = lane & (JW - 1), i0j = lane / JW, istepj = 64 / JW;
  const int kk = lane & (KW - 1), i0k = lane / KW, istepk = 64 / KW;
  // V = I ; finiteness check
  bool ok = true;
  if (jj < D)
    for (int i = i0j; i < D; i += istepj) {
      V[i * LD + jj] = (i == jj) ? 1.0 : 0.0;
      if (!isfinite(A[i * LD + jj])) ok = false;
    }
  ok = __all(ok);
  wave_sync();
  if (!ok) return false;
  if (D == 1) return true;
  const int P = 2 * m;        // players (last one is a bye when D is odd)
  const int rounds = P - 1;
  for (int sweep = 0; sweep < 60; ++sweep) {
    // convergence: off-diagonal mass vs diagonal mass
    double off = 0.0, dia = 0.0;
    if (jj < D)
      for (int i = i0j; i < D; i += istepj) {
        const double a = A[i * LD + jj];
        if (i == jj)
          dia = fma(a, a, dia);
        else
          off = fma(a, a, off);
      }
    for (int s = 32; s > 0; s >>= 1) {
      off += __shfl_xor(off, s);
      dia += __shfl_xor(dia, s);
    }
    if (!(off > 1e-33 * dia)) break;  // also exits on off == 0
    for (int r = 0; r < rounds; ++r) {
      // pair k of round r (circle method)
      if (lane < m) {
        int p, q;
        if (lane == 0) {
          p = P - 1;
          q = r;
        } else {
          p = r + lane;
          if (p >= P - 1) p -= P - 1;
          q = r - lane;
          if (q < 0) q += P - 1;
        }
        if (p > q) {
          const int tmp = p;
          p = q;
          q = tmp;
        }
        double c = 1.0, s = 0.0;
        if (q < D) {
          const double apq = A[p * LD + q];
          if (apq != 0.0) {
            const double app = A[p * LD + p], aqq = A[q * LD + q];
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(fma(tau, tau, 1.0)));
            c = 1.0 / sqrt(fma(t, t, 1.0));
            s = t * c;
          }
        } else {
          p = -1;  // bye
        }
        rc[lane] = c;
        rs[lane] = s;
        rp[lane] = p;
        rp[64 + lane] = q;
      }
      wave_sync();
      // columns: A <- A J, V <- V J      items (i, k)
      if (kk < m) {
        const int p = rp[kk], q = rp[64 + kk];
        if (p >= 0) {
          const double c = rc[kk], s = rs[kk];
          for (int i = i0k; i < D; i += istepk) {
            const double aip = A[i * LD + p], aiq = A[i * LD + q];
            A[i * LD + p] = c * aip - s * aiq;
            A[i * LD + q] = s * aip + c * aiq;
            const double vip = V[i * LD + p], viq = V[i * LD + q];
            V[i * LD + p] = c * vip - s * viq;
            V[i * LD + q] = s * vip + c * viq;
          }
        }
      }
      wave_sync();
      // rows: A <- J^T A                 items (k, j)
      if (jj < D)
        for (int k = i0j; k < m; k += istepj) {
          const int p = rp[k], q = rp[64 + k];
          if (p >= 0) {
            const double c = rc[k], s = rs[k];
            const double apj = A[p * LD + jj], aqj = A[q * LD + jj];
            A[p * LD + jj] = c * apj - s * aqj;
            A[q * LD + jj] = s * apj + c * aqj;
          }
        }
      wave_sync();
      if (lane < m && rp[lane] >= 0) {
        const int p = rp[lane], q = rp[64 + lane];
        A[p * LD + q] = 0.0;
        A[q * LD + p] = 0.0;
      }
      wave_sync();
    }
  }
  return true;
}

// Sort eigenpairs ascending (LAPACK order), fix the sign of every eigenvector so
// its largest-magnitude component is positive (our canonical choice: LAPACK's
// sign is arbitrary).  lam[k], V[:,k] <- sorted; uses AX as scratch.  Wave 0.
static __device__ void sort_eigs_wave(const double* A, double* V, double* lam, int* order, double* scratch,
                               int D, int LD) {
  const int lane = threadIdx.x & 63;
  for (int k = lane; k < D; k += 64) {
    const double mine = A[k * LD + k];
    int rank = 0;
    for (int j = 0; j < D; ++j) {
      const double other = A[j * LD + j];
      if (other < mine || (other == mine && j < k)) ++rank;
    }
    order[rank] = k;  // NaNs compare false everywhere: caught by the caller
  }
  wave_sync();
  for (int e = lane; e < D * D; e += 64) {
    const int i = e / D, k = e % D;
    scratch[i * LD + k] = V[i * LD + order[k]];
  }
  for (int k = lane; k < D; k += 64) lam[k] = A[order[k] * LD + order[k]];
  wave_sync();
  for (int k = lane; k < D; k += 64) {
    double best = 0.0;
    int bi = 0;
    for (int i = 0; i < D; ++i) {
      const double a = fabs(scratch[i * LD + k]);
      if (a > best) {
        best = a;
        bi = i;
      }
    }
    const double sg = scratch[bi * LD + k] < 0.0 ? -1.0 : 1.0;
    for (int i = 0; i < D; ++i) V[i * LD + k] = sg * scratch[i * LD + k];
  }
  wave_sync();
}

// ---- rotation helpers of the one-barrier-per-round Jacobi (rebuild.hip, wide.hip) -----------
static __device__ __forceinline__ double rsqrt_nr(double x) {
  // v_rsq_f64 is good to 2^-24 (measured, tools/micro/rsq_acc.hip); two Newton steps reach 2^-52
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double xy = x * y;
    const double e = fma(-xy, y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  return y;
}

// (c, s) of the Jacobi rotation annihilating apq.  Branch-free; the matrix is pre-scaled
// to max|a_ij| < 1, so d^2 + b^2 cannot overflow, and a pair whose d^2 + b^2 underflows
// is below any convergence threshold and is left alone.
static __device__ __forceinline__ void jacobi_rotation(double app, double aqq, double apq, double& c, double& s) {
  const double d = aqq - app, b = 2.0 * apq;
  double q = fma(d, d, b * b);
  const bool none = (apq == 0.0) || !(q > 0.0);  // also the padding index of odd D
  q = none ? 1.0 : q;
  const double rh = rsqrt_nr(q);
  const double cc = fma(0.5 * fabs(d), rh, 0.5);  // cos^2 in [1/2, 1]
  const double rcc = rsqrt_nr(cc);
  c = none ? 1.0 : cc * rcc;
  s = none ? 0.0 : (d >= 0.0 ? 0.5 : -0.5) * b * rh * rcc;
}

// where the circle method moves the occupant of position `pos` (top row = even
// positions T[k] = 2k, bottom row = odd positions B[k] = 2k+1; T[0] is fixed)
static __device__ __forceinline__ int jacobi_dest(int pos, int m) {
  const int k = pos >> 1;
  if (pos & 1) {
    if (k == 0) return m > 1 ? 2 : 1;  // B[0] -> T[1]
    return 2 * (k - 1) + 1;            // B[k] -> B[k-1]
  }
  if (k == 0) return 0;
  if (k == m - 1) return 2 * (m - 1) + 1;  // T[m-1] -> B[m-1]
  return 2 * (k + 1);                      // T[k] -> T[k+1]
}

}  // namespace dh_eig
