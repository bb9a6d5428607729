// Bounding kernels: ellipsoid membership (K5).  Rebuild kernels follow below.
#include "ctx.h"

using namespace dh;

namespace {


// MultiEllipsoid.within / contains (bounding.py:502-523): lane = candidate point,
// the loop runs over ellipsoids so centre and precision matrix are wave-uniform
// (scalar loads); each ellipsoid's verdict for 64 candidates is one ballot word.
template <int N>
__global__ void __launch_bounds__(64)
    contains_kernel(const double* __restrict__ x, int k, int d, const double* __restrict__ ctrs,
                    const double* __restrict__ ams, int m, int mode, int32_t* count, uint64_t* mask,
                    double* quad) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  const bool live = w < k;
  const int wi = live ? w : k - 1;
  double xx[N];
#pragma unroll
  for (int i = 0; i < N; ++i) xx[i] = (i < d) ? x[(size_t)wi * d + i] : 0.0;
  int cnt = 0;
  const int nwords = (k + 63) / 64;
  for (int a = 0; a < m; ++a) {
    const double* __restrict__ c = ctrs + (size_t)a * d;
    const double* __restrict__ A = ams + (size_t)a * d * d;
    double dl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dl[i] = (i < d) ? xx[i] - c[i] : 0.0;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i < d) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (j < d) r = fma(A[i * d + j], dl[j], r);
        q = fma(dl[i], r, q);
      }
    }
    const bool in = live && (mode == 0 ? (q < 1.0) : (sqrt(q) <= 1.0));
    cnt += in ? 1 : 0;
    const unsigned long long b = __ballot(in);
    if (mask && threadIdx.x == 0) mask[(size_t)a * nwords + blockIdx.x] = b;
    if (quad && live) quad[(size_t)w * m + a] = q;
  }
  if (live) count[w] = cnt;
}

// Sampler.propose_live's membership test of the start points (sampler.py:484-489) for an ensemble: candidate w
// belongs to run w / wpr and is tested against THAT run's ellipsoids (bound arrays strided by max_ells); a run
// with a candidate outside its bound gets flag[run] = 1 (the caller rebuilds that run's bound).  Same lane map
// and arithmetic as contains_kernel; the runs of a wavefront are served one after the other so that centre
// and precision matrix stay wave-uniform.  strict = 1: MultiEllipsoid.contains (any quad < 1), 0:
// Ellipsoid.contains (sqrt(quad) <= 1).
template <int N>
__global__ void __launch_bounds__(64)
    contains_runs_kernel(const double* __restrict__ x, int k, int d, int wpr, const double* __restrict__ ctrs,
                         const double* __restrict__ ams, const int* __restrict__ nells, int max_ells, int strict,
                         const int* __restrict__ run_mode, int my_mode, const int* __restrict__ bstatus, int* flag,
                         int* first) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  const bool live = w < k;
  const int wi = live ? w : k - 1;
  const int my_run = wi / wpr;
  double xx[N];
#pragma unroll
  for (int i = 0; i < N; ++i) xx[i] = (i < d) ? x[(size_t)wi * d + i] : 0.0;
  bool pending = live;
  while (__any(pending)) {
    // the lowest pending lane's run, for the whole wave (wave-uniform operands below)
    const unsigned long long pm = __ballot(pending);
    // (readfirstlane: the compiler must KNOW the run is wave-uniform, else centre and matrix are fetched per lane:
    // 625 vector loads of one address each instead of scalar loads -- 39 us per 32 768 25-D points)
    const int run = __builtin_amdgcn_readfirstlane(__shfl(my_run, __ffsll((long long)pm) - 1));
    const bool mine = pending && my_run == run;
    const bool served = (!run_mode || run_mode[run] == my_mode) && (!bstatus || bstatus[run] == 0);
    if (served) {
      const int m = nells ? nells[run] : 1;
      bool inside = false;
      for (int a = 0; a < m; ++a) {
        const double* __restrict__ c = ctrs + ((size_t)run * max_ells + a) * d;
        const double* __restrict__ A = ams + ((size_t)run * max_ells + a) * d * d;
        double dl[N];
#pragma unroll
        for (int i = 0; i < N; ++i) dl[i] = (i < d) ? xx[i] - c[i] : 0.0;
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          if (i < d) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j)
              if (j < d) r = fma(A[i * d + j], dl[j], r);
            q = fma(dl[i], r, q);
          }
        }
        inside = inside || (strict ? (q < 1.0) : (sqrt(q) <= 1.0));
      }
      const unsigned long long outm = __ballot(mine && !inside);
      if (outm && threadIdx.x == (unsigned)(__ffsll((long long)outm) - 1)) {
        atomicOr(&flag[run], 1);
        if (first) atomicMin(&first[run], w - run * wpr);  // the earliest queue entry outside (lanes ascend with w)
      }
    }
    if (mine) pending = false;
  }
}

// The same test with the ellipsoid's centre and precision matrix staged in LDS (round 6), for queues whose length is a
// multiple of 256 (a workgroup's 256 candidates are of ONE run).  The form above fetches the matrix through the scalar
// cache: 625 doubles per ellipsoid and wavefront at D = 25, eighty dependent 64-byte scalar loads in front of a chain of
// 625 multiply-adds -- 26 us per 64 x 512 candidates, 4 % of the C2 loop.  Here the workgroup loads the 5 KB once
// (coalesced) and every lane reads its operands as LDS broadcasts.  The same multiply-adds in the same order: the same
// verdicts.
template <int N>
__global__ void __launch_bounds__(256)
    contains_runs_lds_kernel(const double* __restrict__ x, int k, int d, int wpr, const double* __restrict__ ctrs,
                             const double* __restrict__ ams, const int* __restrict__ nells, int max_ells, int strict,
                             const int* __restrict__ run_mode, int my_mode, const int* __restrict__ bstatus, int* flag,
                             int* first) {
  __shared__ double sA[N * N + N];
  const int t = threadIdx.x, w = blockIdx.x * 256 + t;
  const bool live = w < k;
  const int wi = live ? w : k - 1;
  const int run = (blockIdx.x * 256) / wpr;  // (wpr % 256 == 0: the same for the whole workgroup)
  const bool served = (!run_mode || run_mode[run] == my_mode) && (!bstatus || bstatus[run] == 0);
  if (!served) return;
  double xx[N];
#pragma unroll
  for (int i = 0; i < N; ++i) xx[i] = (i < d) ? x[(size_t)wi * d + i] : 0.0;
  const int m = nells ? nells[run] : 1;
  bool inside = false;
  for (int a = 0; a < m; ++a) {
    const double* __restrict__ c = ctrs + ((size_t)run * max_ells + a) * d;
    const double* __restrict__ A = ams + ((size_t)run * max_ells + a) * d * d;
    if (a) __syncthreads();
    for (int e = t; e < d * d + d; e += 256) sA[e] = e < d * d ? A[e] : c[e - d * d];
    __syncthreads();
    double dl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) dl[i] = (i < d) ? xx[i] - sA[d * d + (i < d ? i : 0)] : 0.0;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i < d) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j)
          if (j < d) r = fma(sA[i * d + j], dl[j], r);
        q = fma(dl[i], r, q);
      }
    }
    inside = inside || (strict ? (q < 1.0) : (sqrt(q) <= 1.0));
  }
  const bool out = live && !inside;
  if (out) {
    atomicOr(&flag[run], 1);
    if (first) atomicMin(&first[run], w - run * wpr);
  }
}

// The same test above the register-resident dimensions.  One wavefront per start point re-read the run's precision
// matrix (320 KB at D = 200) for every point: 109 us per 2 048-walker fill of C4, L2-bandwidth bound.  Now a workgroup of
// four wavefronts takes P <= 8 start points OF ONE RUN: the points' offsets from the centre sit in LDS, wavefront w
// forms r_i = sum_j A[j][i] dl[j] for the rows i = l + 64 c of its chunks c = w, w + 4 (consecutive lanes read
// consecutive doubles; a matrix element is loaded once and serves all P points), and wavefront 0 folds
// q = sum_i dl_i r_i exactly as the one-wavefront form did -- per lane over its chunks in ascending order, then
// across the lanes -- so every point's verdict comes from the same arithmetic.
constexpr int kWideBlock = 8;
__global__ void __launch_bounds__(256)
    contains_runs_wide_kernel(const double* __restrict__ x, int k, int d, int wpr, int P, const double* __restrict__ ctrs,
                              const double* __restrict__ ams, const int* __restrict__ nells, int max_ells, int strict,
                              const int* __restrict__ run_mode, int my_mode, const int* __restrict__ bstatus, int* flag,
                              int* first) {
  extern __shared__ double sm[];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int C = (d + 63) >> 6;           // 64-row chunks
  double* dl = sm;                       // P x d
  double* rb = sm + (size_t)P * d;       // C x P x 64: r of chunk c, point p, lane l
  __shared__ unsigned inside_s;
  const int bpr = (wpr + P - 1) / P;  // blocks per run
  const int run = blockIdx.x / bpr, w0 = run * wpr + (blockIdx.x % bpr) * P;
  if (w0 >= k) return;
  const int np = min(P, min(run * wpr + wpr, k) - w0);
  if ((run_mode && run_mode[run] != my_mode) || (bstatus && bstatus[run] != 0)) return;
  const int m = nells ? nells[run] : 1;
  const unsigned all = (1u << np) - 1u;
  unsigned inside = 0;  // bit p: point p lies in some ellipsoid (workgroup-uniform)
  for (int a = 0; a < m && inside != all; ++a) {
    const double* __restrict__ c = ctrs + ((size_t)run * max_ells + a) * d;
    const double* __restrict__ A = ams + ((size_t)run * max_ells + a) * d * d;
    __syncthreads();
    for (int e = t; e < np * d; e += 256) {
      const int p = e / d, i = e - p * d;
      dl[e] = x[(size_t)(w0 + p) * d + i] - c[i];
    }
    __syncthreads();
    for (int ch = wv; ch < C; ch += 4) {
      const int i = ch * 64 + lane;
      double r[kWideBlock];
#pragma unroll
      for (int p = 0; p < kWideBlock; ++p) r[p] = 0.0;
      if (i < d)
        for (int j = 0; j < d; ++j) {
          const double aji = A[(size_t)j * d + i];
#pragma unroll
          for (int p = 0; p < kWideBlock; ++p)
            if (p < np) r[p] = fma(aji, dl[p * d + j], r[p]);
        }
#pragma unroll
      for (int p = 0; p < kWideBlock; ++p)
        if (p < np) rb[((size_t)ch * P + p) * 64 + lane] = r[p];
    }
    __syncthreads();
    if (wv == 0) {
      unsigned got = 0;
      for (int p = 0; p < np; ++p) {
        double q = 0.0;
        for (int ch = 0; ch < C; ++ch) {
          const int i = ch * 64 + lane;
          if (i < d) q = fma(dl[p * d + i], rb[((size_t)ch * P + p) * 64 + lane], q);
        }
        for (int sft = 32; sft > 0; sft >>= 1) q += __shfl_xor(q, sft);
        if (strict ? (q < 1.0) : (sqrt(q) <= 1.0)) got |= 1u << p;
      }
      if (lane == 0) inside_s = inside | got;
    }
    __syncthreads();
    inside = inside_s;
  }
  const unsigned out = all & ~inside;
  if (out && t == 0) {
    atomicOr(&flag[run], 1);
    if (first) atomicMin(&first[run], w0 + (__ffs((int)out) - 1) - run * wpr);
  }
}

}  // namespace

namespace dh {
int contains_runs_launch(dh_ctx* ctx, const double* x, int k, int d, int wpr, const double* ctrs, const double* ams,
                         const int* nells, int max_ells, int strict, const int* run_mode, int my_mode,
                         const int* bstatus, int* flag, int* first) {
  if (k <= 0) return DH_OK;
  if (d > kMaxRegDim) {
    const int P = d <= 256 ? kWideBlock : 4;  // (LDS: P d + ceil(d / 64) P 64 doubles <= 32 KB)
    const int bpr = (wpr + P - 1) / P, nruns = (k + wpr - 1) / wpr;
    const size_t lds = ((size_t)P * d + (size_t)((d + 63) / 64) * P * 64) * 8;
    hipLaunchKernelGGL(contains_runs_wide_kernel, dim3(nruns * bpr), dim3(256), lds, ctx->stream, x, k, d, wpr, P, ctrs, ams,
                       nells, max_ells, strict, run_mode, my_mode, bstatus, flag, first);
    return hip_ok(ctx, hipGetLastError(), "contains_runs launch") ? DH_OK : DH_ERR_HIP;
  }
  const bool lds_form = !(getenv("DH_CONTAINS_LDS") && atoi(getenv("DH_CONTAINS_LDS")) == 0);
  if (lds_form && d >= 9 && wpr % 256 == 0) {
    bool hit2 = false;
#define X(NN)                                                                                                      \
  if (!hit2 && d <= NN) {                                                                                          \
    hit2 = true;                                                                                                   \
    hipLaunchKernelGGL(contains_runs_lds_kernel<NN>, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, x, k, d, wpr, ctrs, \
                       ams, nells, max_ells, strict, run_mode, my_mode, bstatus, flag, first);                      \
  }
    DH_DIM_LIST(X)
#undef X
    return hip_ok(ctx, hipGetLastError(), "contains_runs launch") ? DH_OK : DH_ERR_HIP;
  }
  const dim3 grid((k + 63) / 64), block(64);
  bool hit = false;
#define X(NN)                                                                                              \
  if (!hit && d <= NN) {                                                                                   \
    hit = true;                                                                                            \
    hipLaunchKernelGGL(contains_runs_kernel<NN>, grid, block, 0, ctx->stream, x, k, d, wpr, ctrs, ams, nells, \
                       max_ells, strict, run_mode, my_mode, bstatus, flag, first);                         \
  }
  DH_DIM_LIST(X)
#undef X
  return hip_ok(ctx, hipGetLastError(), "contains_runs launch") ? DH_OK : DH_ERR_HIP;
}
}  // namespace dh

extern "C" {

int dh_contains(dh_ctx* ctx, const double* x, int k, int d, const double* ctrs, const double* ams,
                int m, int mode, int32_t* count, uint64_t* mask, double* quad) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!x || !ctrs || !ams || !count || d < 1 || m < 1 || (mode != 0 && mode != 1))
    return fail(ctx, DH_ERR_ARG, "contains: bad arguments (d=%d m=%d mode=%d)", d, m, mode);
  arena_reset(ctx);
  const size_t nwords = (size_t)(k + 63) / 64;
  int rc = arena_reserve(ctx, ((size_t)k * d + (size_t)m * d + (size_t)m * d * d + (size_t)k * m) * 8 +
                                  (size_t)k * 4 + nwords * m * 8 + 8192);
  if (rc) return rc;
  const double* d_x = arena_up(ctx, x, (size_t)k * d);
  const double* d_c = arena_up(ctx, ctrs, (size_t)m * d);
  const double* d_a = arena_up(ctx, ams, (size_t)m * d * d);
  int32_t* d_cnt = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_mask = mask ? (uint64_t*)arena_get(ctx, nwords * m * 8) : nullptr;
  double* d_q = quad ? (double*)arena_get(ctx, (size_t)k * m * 8) : nullptr;
  if (!d_x || !d_c || !d_a || !d_cnt || (mask && !d_mask) || (quad && !d_q)) return DH_ERR_NOMEM;
  const dim3 grid((k + 63) / 64), block(64);
  bool hit = false;
  if (d > kMaxRegDim) {
    hit = true;
    rc = wide_contains_launch(ctx, d_x, k, d, d_c, d_a, m, mode, d_cnt, d_mask, d_q);
    if (rc) return rc;
  }
#define X(NN)                                                                                   \
  if (!hit && d <= NN) {                                                                        \
    hit = true;                                                                                 \
    hipLaunchKernelGGL(contains_kernel<NN>, grid, block, 0, ctx->stream, d_x, k, d, d_c, d_a, m, \
                       mode, d_cnt, d_mask, d_q);                                               \
  }
  DH_DIM_LIST(X)
#undef X
  if (!hip_ok(ctx, hipGetLastError(), "contains launch")) return DH_ERR_HIP;
  if (!down(ctx, count, d_cnt, (size_t)k) || !down(ctx, mask, d_mask, nwords * m) ||
      !down(ctx, quad, d_q, (size_t)k * m))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"
