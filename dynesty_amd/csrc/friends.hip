// RadFriends / SupFriends bounds: N-balls / N-cubes of one common shape
// centred on every live point.  Reference: bounding.py:734-1263 (classes), :1651-1702
// (radius helpers).  Everything here is brute force over the N centres -- the natural
// shape for the device: the reference's KD-tree / pdist / single-linkage calls become
//
//   fr_matmul        Y = X M                      points into the whitened frame of a metric
//   fr_adjacency     bit matrix [i][j/64] of |y_i - y_j|_2 <= 1      (single linkage cut at 1
//   fr_components    = connected components: min-label hooking + pointer jumping)
//   fr_cluster_mean  per-component mean, fixed order      fr_recentre / fr_colmean / fr_cov
//   fr_shape         eigh(cov) by one wavefront -> sqrtm, pinvh(cov), pinvh(sqrtm)
//   fr_nn            leave-one-out / bootstrap nearest-neighbour distance (2- or max-norm)
//   fr_within        membership counts + ballot bit rows of candidate points
//   fr_draw          Bound.sample(s) from ONE generator: draws by lane 0, overlap count by the wave
#include <math.h>

#include "ctx.h"
#include "eig_wave.h"
#include "rng_pcg64.h"

using namespace dh;

namespace {

constexpr int kT = 256;
enum : int { KIND_BALLS = 0, KIND_CUBES = 1 };

// ---- Y = X M  (n x d times d x d), one thread per output --------------------------------
__global__ void __launch_bounds__(kT) fr_matmul(const double* __restrict__ X, const double* __restrict__ M,
                                                 int n, int d, double* __restrict__ Y) {
  extern __shared__ double sm[];  // M, d*d
  for (int e = threadIdx.x; e < d * d; e += kT) sm[e] = M[e];
  __syncthreads();
  const long long o = (long long)blockIdx.x * kT + threadIdx.x;
  if (o >= (long long)n * d) return;
  const int i = (int)(o / d), j = (int)(o - (long long)i * d);
  const double* x = X + (size_t)i * d;
  double s = 0.0;
  for (int k = 0; k < d; ++k) s = fma(x[k], sm[k * d + j], s);
  Y[o] = s;
}

// ---- adjacency bits: row i, word w holds j = 64 w + lane with d_M(x_i, x_j) <= 1 ------------
// By construction of the radius, the point with the largest nearest-neighbour distance sits
// EXACTLY on the linkage threshold in the metric of the previous update (d = 1 in exact
// arithmetic), so whether it joins its neighbour's cluster is decided by the last bit of the
// distance.  To take the reference's decision, the distance is evaluated with the very
// arithmetic of scipy's pdist(metric='mahalanobis') (distance_impl.h): delta = u - v,
// t_a = sum_b delta_b VI[a][b] (b ascending), s = sum_a delta_a t_a (a ascending), sqrt(s) --
// plain multiply-add, no FMA contraction.  O(d^2) per pair; n^2 d^2 is still small (2.5 Gflop
// at n = 2000, d = 25).  grid (ceil(n/64) word columns, ceil(n/4) row groups); a wave owns row
// i, lane = j.
#pragma clang fp contract(off)
__global__ void __launch_bounds__(kT) fr_adjacency(const double* __restrict__ X, const double* __restrict__ VI,
                                                    int n, int d, unsigned long long* __restrict__ bits, int nw) {
  extern __shared__ double sm[];  // VI d*d | per wave: u d | columns [d][64]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.y * 4 + wv, w = blockIdx.x;
  double* vi = sm;
  double* u = sm + d * d + wv * (d + d * 64);
  double* col = u + d;
  for (int e = threadIdx.x; e < d * d; e += kT) vi[e] = VI[e];
  const int j = w * 64 + lane;
  const int ii = i < n ? i : n - 1, jj = j < n ? j : n - 1;
  for (int k = lane; k < d; k += 64) u[k] = X[(size_t)ii * d + k];
  __syncthreads();
  // delta = u - v, kept as this lane's LDS column
  for (int k = 0; k < d; ++k) col[k * 64 + lane] = u[k] - X[(size_t)jj * d + k];
  double s = 0.0;
  for (int a = 0; a < d; ++a) {
    const double* row = vi + a * d;
    double t = 0.0;
    for (int b = 0; b < d; ++b) t = t + col[b * 64 + lane] * row[b];
    s = s + col[a * 64 + lane] * t;
  }
  const bool adj = (i < n) && (j < n) && (sqrt(s) <= 1.0);
  const unsigned long long m = __ballot(adj);
  if (lane == 0 && i < n) bits[(size_t)i * nw + w] = m;
}
#pragma clang fp contract(fast)

// ---- connected components of the bit graph: labels -> smallest index of the component -----
// One workgroup.  Round: hook (every vertex takes the smallest label among its neighbours and
// pushes it to its current root), then pointer jumping until stable; repeat until a round
// changes nothing.  The fixed point is order-independent.
__global__ void __launch_bounds__(1024) fr_components(const unsigned long long* __restrict__ bits, int n, int nw,
                                                       int* __restrict__ label, int* __restrict__ ncomp) {
  const int t = threadIdx.x;
  __shared__ int changed;
  for (int i = t; i < n; i += 1024) label[i] = i;
  __syncthreads();
  for (int round = 0; round < 4 * 64; ++round) {
    if (t == 0) changed = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
      int m = label[i];
      const unsigned long long* row = bits + (size_t)i * nw;
      for (int w = 0; w < nw; ++w) {
        unsigned long long b = row[w];
        while (b) {
          const int j = w * 64 + __ffsll((long long)b) - 1;
          b &= b - 1;
          const int lj = label[j];
          if (lj < m) m = lj;
        }
      }
      if (m < label[i]) {
        atomicMin(&label[label[i]], m);
        atomicMin(&label[i], m);
        changed = 1;
      }
    }
    __syncthreads();
    // pointer jumping
    for (;;) {
      __shared__ int moved;
      if (t == 0) moved = 0;
      __syncthreads();
      for (int i = t; i < n; i += 1024) {
        const int l = label[i], ll = label[l];
        if (ll < l) {
          label[i] = ll;
          moved = 1;
        }
      }
      __syncthreads();
      const int mv = moved;
      __syncthreads();
      if (!mv) break;
    }
    const int ch = changed;
    __syncthreads();
    if (!ch) break;
  }
  int c = 0;
  for (int i = t; i < n; i += 1024) c += label[i] == i ? 1 : 0;
  __shared__ int tot;
  if (t == 0) tot = 0;
  __syncthreads();
  atomicAdd(&tot, c);
  __syncthreads();
  if (t == 0) *ncomp = tot;
}

// ---- mean of every component (block = candidate root), members visited in index order -------
__global__ void __launch_bounds__(64) fr_cluster_mean(const double* __restrict__ X, const int* __restrict__ label,
                                                       int n, int d, double* __restrict__ mean) {
  const int r = blockIdx.x;
  if (label[r] != r) return;
  for (int k = threadIdx.x; k < d; k += 64) {
    double s = 0.0;
    int c = 0;
    for (int i = r; i < n; ++i)
      if (label[i] == r) {
        s += X[(size_t)i * d + k];
        ++c;
      }
    mean[(size_t)r * d + k] = s / (double)c;
  }
}

__global__ void __launch_bounds__(kT) fr_recentre(const double* __restrict__ X, const int* __restrict__ label,
                                                   const double* __restrict__ mean, int n, int d,
                                                   double* __restrict__ out) {
  const long long o = (long long)blockIdx.x * kT + threadIdx.x;
  if (o >= (long long)n * d) return;
  const int i = (int)(o / d), k = (int)(o - (long long)i * d);
  out[o] = X[o] - mean[(size_t)label[i] * d + k];
}

// deterministic block sum
__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < kT / 64; ++i) r += red[i];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(kT) fr_colmean(const double* __restrict__ X, int n, int d, double* __restrict__ m) {
  __shared__ double red[kT / 64];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kT) s += X[(size_t)i * d + k];
  s = block_sum(s, red);
  if (threadIdx.x == 0) m[k] = s / (double)n;
}

// np.cov(X, rowvar=False): entry (a, b), ddof = 1
__global__ void __launch_bounds__(kT) fr_cov(const double* __restrict__ X, const double* __restrict__ m, int n, int d,
                                              double* __restrict__ cov) {
  __shared__ double red[kT / 64];
  const int a = blockIdx.x / d, b = blockIdx.x % d;
  if (b < a) return;
  const double ma = m[a], mb = m[b];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kT) s = fma(X[(size_t)i * d + a] - ma, X[(size_t)i * d + b] - mb, s);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const double c = s / (double)(n - 1);
    cov[a * d + b] = c;
    cov[b * d + a] = c;
  }
}

// ---- eigh(cov) -> am = pinvh(cov), axes = sqrtm(cov), axes_inv = pinvh(axes), sum log lambda ----
// One wavefront; matrices in LDS.  pinvh keeps |lambda| > d * eps * max|lambda| (scipy default).
__global__ void __launch_bounds__(64) fr_shape(const double* __restrict__ cov, int d, double* __restrict__ am,
                                                double* __restrict__ axes, double* __restrict__ axes_inv,
                                                double* __restrict__ info /* [0] sum log lam, [1] n dropped */,
                                                int* __restrict__ status) {
  extern __shared__ double sm[];
  const int LD = d | 1, lane = threadIdx.x;
  double* A = sm;
  double* V = A + d * LD;
  double* S = V + d * LD;
  double* lam = S + d * LD;
  double* rc = lam + d;
  double* rs = rc + 64;
  int* ri = (int*)(rs + 64);
  int* order = ri + 128;
  for (int e = lane; e < d * d; e += 64) A[(e / d) * LD + e % d] = cov[e];
  dh_eig::wave_sync();
  const bool fin = dh_eig::jacobi_wave(A, V, d, LD, rc, rs, ri);
  if (!fin) {
    if (lane == 0) *status = DH_ERR_VALUE;
    return;
  }
  dh_eig::sort_eigs_wave(A, V, lam, order, S, d, LD);
  double top = 0.0;
  for (int k = 0; k < d; ++k) top = fmax(top, fabs(lam[k]));
  const double eps = 2.220446049250313e-16;
  const double cut = (double)d * eps * top, cut_s = (double)d * eps * sqrt(top);
  double slog = 0.0;
  int dropped = 0;
  for (int k = 0; k < d; ++k) {
    if (lam[k] > cut)
      slog += log(lam[k]);
    else
      ++dropped;
  }
  for (int e = lane; e < d * d; e += 64) {
    const int i = e / d, j = e % d;
    double sa = 0.0, sx = 0.0, si = 0.0;
    for (int k = 0; k < d; ++k) {
      const double l = lam[k], vv = V[i * LD + k] * V[j * LD + k];
      const double rt = l > 0.0 ? sqrt(l) : 0.0;
      if (fabs(l) > cut) sa = fma(vv, 1.0 / l, sa);
      sx = fma(vv, rt, sx);
      if (rt > cut_s) si = fma(vv, 1.0 / rt, si);
    }
    am[e] = sa;
    axes[e] = sx;
    axes_inv[e] = si;
  }
  if (lane == 0) {
    info[0] = slog;
    info[1] = (double)dropped;
    *status = DH_OK;
  }
}

// ---- nearest-neighbour distance in the whitened frame -------------------------------------------
// grid (ceil(n/kT), replicas).  Query i (skipped when in_mask says it was resampled), candidates j
// (only resampled ones under a mask; j != i without one).  Output nn[b][i] (or -1 for skipped).
__global__ void __launch_bounds__(kT) fr_nn(const double* __restrict__ Y, int n, int d, int kind,
                                             const unsigned char* __restrict__ in_mask, double* __restrict__ nn) {
  extern __shared__ double tile[];  // 64 x d candidate rows
  const int b = blockIdx.y, t = threadIdx.x;
  const int i = blockIdx.x * kT + t;
  const unsigned char* mk = in_mask ? in_mask + (size_t)b * n : nullptr;
  const bool active = i < n && (!mk || !mk[i]);
  const double* yi = Y + (size_t)(i < n ? i : 0) * d;
  double best = INFINITY;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int cnt = min(64, n - j0);
    __syncthreads();
    for (int e = t; e < cnt * d; e += kT) tile[e] = Y[(size_t)j0 * d + e];
    __syncthreads();
    if (!active) continue;
    for (int jj = 0; jj < cnt; ++jj) {
      const int j = j0 + jj;
      if (mk ? !mk[j] : j == i) continue;
      const double* yj = tile + jj * d;
      double s = 0.0;
      if (kind == KIND_BALLS) {
        for (int k = 0; k < d; ++k) {
          const double e = yi[k] - yj[k];
          s = fma(e, e, s);
        }
      } else {
        for (int k = 0; k < d; ++k) s = fmax(s, fabs(yi[k] - yj[k]));
      }
      best = fmin(best, s);
    }
  }
  if (i < n) nn[(size_t)b * n + i] = active ? (kind == KIND_BALLS ? sqrt(best) : best) : -1.0;
}

__global__ void __launch_bounds__(kT) fr_max(const double* __restrict__ v, long long n, double* __restrict__ out) {
  __shared__ double red[kT / 64];
  double m = -INFINITY;
  for (long long i = threadIdx.x; i < n; i += kT) m = fmax(m, v[i]);
  for (int s = 32; s > 0; s >>= 1) m = fmax(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kT / 64; ++i) m = fmax(m, red[i]);
    *out = m;
  }
}

// ---- membership: lane = candidate, loop over the centres (whitened, wave-uniform rows) ------------
// counts[c] = number of balls / cubes containing candidate c; bits[c][j/64] optional.
__global__ void __launch_bounds__(64) fr_within(const double* __restrict__ CT /* n x d centres, whitened */,
                                                 const double* __restrict__ XT /* m x d candidates, whitened */,
                                                 int n, int m, int d, int kind, int* __restrict__ counts,
                                                 unsigned long long* __restrict__ bits, int nw) {
  extern __shared__ double xs[];  // [d][64] candidate columns
  const int lane = threadIdx.x, c = blockIdx.x * 64 + lane;
  for (int k = 0; k < d; ++k) xs[k * 64 + lane] = c < m ? XT[(size_t)c * d + k] : 0.0;
  int cnt = 0;
  unsigned long long word = 0;
  for (int j = 0; j < n; ++j) {
    const double* cj = CT + (size_t)j * d;
    double s = 0.0;
    if (kind == KIND_BALLS) {
      for (int k = 0; k < d; ++k) {
        const double e = cj[k] - xs[k * 64 + lane];
        s = fma(e, e, s);
      }
      s = sqrt(s);
    } else {
      for (int k = 0; k < d; ++k) s = fmax(s, fabs(cj[k] - xs[k * 64 + lane]));
    }
    const bool in = s <= 1.0;
    cnt += in ? 1 : 0;
    if (bits) {
      if (in) word |= 1ull << (j & 63);
      if ((j & 63) == 63 || j == n - 1) {
        if (c < m) bits[(size_t)c * nw + (j >> 6)] = word;
        word = 0;
      }
    }
  }
  if (c < m) counts[c] = cnt;
}

// ---- Bound.sample / samples from ONE generator (bounding.py:795-847, 1066-1117) ----------------
// Sequential by construction.  Lane 0 draws; the overlap count over the n centres is shared by
// the wave.  state: 6 words {state hi, lo, inc hi, lo, has_uint32, uinteger}.
__global__ void __launch_bounds__(64) fr_draw(const uint64_t* state_in, int nsamp, int n, int d, int kind,
                                               const double* __restrict__ ctrs, const double* __restrict__ CT,
                                               const double* __restrict__ axes, const double* __restrict__ axes_inv,
                                               int return_q, double* xs, int32_t* qs, uint64_t* state_out,
                                               const uint64_t* zki, const uint64_t* zwi, const uint64_t* zfi) {
  __shared__ ZigLds zig;
  extern __shared__ double wk[];  // ds d | x d | y d
  zig_stage(&zig, zki, zwi, zfi);
  const int lane = threadIdx.x;
  double* ds = wk;
  double* x = wk + d;
  double* y = wk + 2 * d;
  __shared__ int sh_take;
  Pcg64 g;
  if (lane == 0) {
    g.load(state_in);
    g.has32 = (uint32_t)state_in[4];
    g.buf32 = (uint32_t)state_in[5];
  }
  for (int s = 0; s < nsamp; ++s) {
    for (;;) {
      if (lane == 0) {
        if (kind == KIND_BALLS) {
          double ss = 0.0;
          for (int i = 0; i < d; ++i) {
            ds[i] = std_normal(g, &zig);
            ss = fma(ds[i], ds[i], ss);
          }
          const double fac = pow(g.next_double(), 1.0 / (double)d) / sqrt(ss);
          for (int i = 0; i < d; ++i) ds[i] *= fac;
        } else {
          for (int i = 0; i < d; ++i) ds[i] = -1.0 + 2.0 * g.next_double();
        }
        int idx = 0;
        if (n > 1) idx = (int)g.bounded_lemire32((uint32_t)(n - 1));
        // dx = ds . axes ; x = ctr + dx ; y = x . axes_inv
        for (int j = 0; j < d; ++j) {
          double r = 0.0;
          for (int i = 0; i < d; ++i) r = fma(ds[i], axes[i * d + j], r);
          x[j] = ctrs[(size_t)idx * d + j] + r;
        }
        for (int j = 0; j < d; ++j) {
          double r = 0.0;
          for (int i = 0; i < d; ++i) r = fma(x[i], axes_inv[i * d + j], r);
          y[j] = r;
        }
      }
      __syncthreads();
      int q = 1;
      if (n > 1) {
        int c = 0;
        for (int j = lane; j < n; j += 64) {
          const double* cj = CT + (size_t)j * d;
          double sdist = 0.0;
          if (kind == KIND_BALLS) {
            for (int k = 0; k < d; ++k) {
              const double e = cj[k] - y[k];
              sdist = fma(e, e, sdist);
            }
            sdist = sqrt(sdist);
          } else {
            for (int k = 0; k < d; ++k) sdist = fmax(sdist, fabs(cj[k] - y[k]));
          }
          c += sdist <= 1.0 ? 1 : 0;
        }
        for (int sft = 32; sft > 0; sft >>= 1) c += __shfl_xor(c, sft);
        q = c;
      }
      if (lane == 0) {
        bool take = (q == 1) || return_q;
        if (!take && q > 0) take = g.next_double() < (1.0 / (double)q);
        // q == 0 (rounding put the draw outside its own shape): the reference divides by zero; redraw
        sh_take = take ? 1 : 0;
        if (take) {
          for (int i = 0; i < d; ++i) xs[(size_t)s * d + i] = x[i];
          qs[s] = q;
        }
      }
      __syncthreads();
      if (sh_take) break;
    }
  }
  if (lane == 0) {
    g.store(state_out);
    state_out[4] = g.has32;
    state_out[5] = g.buf32;
  }
}

int launch_ok(dh_ctx* ctx, const char* what) {
  return hip_ok(ctx, hipGetLastError(), what) ? DH_OK : DH_ERR_HIP;
}

}  // namespace

int dh::friends_whiten_launch(dh_ctx* ctx, const double* X, const double* M, int n, int d, double* Y) {
  const size_t nd = (size_t)n * d;
  hipLaunchKernelGGL(fr_matmul, dim3((int)((nd + kT - 1) / kT)), dim3(kT), (size_t)d * d * 8, ctx->stream, X, M, n, d, Y);
  return hip_ok(ctx, hipGetLastError(), "friends whiten launch") ? DH_OK : DH_ERR_HIP;
}

extern "C" {

// see include/dynhip.h
int dh_friends_update(dh_ctx* ctx, const double* pts, int n, int d, int kind, const double* am_prev,
                      int nboot, const uint8_t* in_mask, double* cov, double* am, double* axes,
                      double* axes_inv, double* logvol, double* rmax, int32_t* nclusters) {
  DH_CHECK_CTX(ctx);
  if (!pts || n < 2 || d < 1 || d > 64 || (kind != KIND_BALLS && kind != KIND_CUBES) || !cov || !am || !axes ||
      !axes_inv || !logvol || !rmax || nboot < 0 || (nboot > 0 && !in_mask))
    return fail(ctx, DH_ERR_ARG, "friends_update: bad arguments (2 <= n, 1 <= d <= 64)");
  (void)hipSetDevice(ctx->device);
  arena_reset(ctx);
  const size_t nd = (size_t)n * d, dd = (size_t)d * d;
  const int nw = (n + 63) / 64;
  const int reps = nboot > 0 ? nboot : 1;
  const size_t need = nd * 8 * 3 + (size_t)n * nw * 8 + (size_t)n * 4 + dd * 8 * 6 + (size_t)reps * n * 9 + 65536;
  int rc = arena_reserve(ctx, need);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  double* d_x = arena_up(ctx, pts, nd);
  double* d_y = (double*)arena_get(ctx, nd * 8);
  double* d_mv = (double*)arena_get(ctx, nd * 8);  // cluster means (n x d, rows of roots), then recentred points
  double* d_prev = am_prev ? arena_up(ctx, am_prev, dd) : nullptr;
  unsigned long long* d_bits = (unsigned long long*)arena_get(ctx, (size_t)n * nw * 8);
  int* d_label = (int*)arena_get(ctx, (size_t)n * 4);
  int* d_nc = (int*)arena_get(ctx, 8);
  double* d_m = (double*)arena_get(ctx, (size_t)d * 8);
  double* d_cov = (double*)arena_get(ctx, dd * 8);
  double* d_am = (double*)arena_get(ctx, dd * 8);
  double* d_ax = (double*)arena_get(ctx, dd * 8);
  double* d_ai = (double*)arena_get(ctx, dd * 8);
  double* d_info = (double*)arena_get(ctx, 32);
  int* d_st = (int*)arena_get(ctx, 8);
  unsigned char* d_mask = nboot > 0 ? arena_up(ctx, in_mask, (size_t)nboot * n) : nullptr;
  double* d_nn = (double*)arena_get(ctx, (size_t)reps * n * 8);
  double* d_r = (double*)arena_get(ctx, 8);
  if (!d_x || !d_y || !d_mv || !d_bits || !d_label || !d_nc || !d_m || !d_cov || !d_am || !d_ax || !d_ai ||
      !d_info || !d_st || !d_nn || !d_r || (am_prev && !d_prev) || (nboot > 0 && !d_mask))
    return DH_ERR_NOMEM;
  const int gb = (int)((nd + kT - 1) / kT);
  const double* d_src = d_x;  // points whose covariance is taken
  int ncl = 1;
  if (d_prev) {
    // clusters: single linkage cut at Mahalanobis distance 1 in the previous metric
    const size_t lds_adj = (dd + 4 * ((size_t)d + (size_t)d * 64)) * 8;
    if (lds_adj > 159 * 1024) return fail(ctx, DH_ERR_ARG, "friends_update: clustering needs d <= 60 (d = %d)", d);
    DH_DEV_MEMO(attr_adj);
    if (lds_adj > attr_adj) {
      if (!hip_ok(ctx, hipFuncSetAttribute((const void*)fr_adjacency, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_adj), "hipFuncSetAttribute(fr_adjacency)"))
        return DH_ERR_HIP;
      attr_adj = lds_adj;
    }
    hipLaunchKernelGGL(fr_adjacency, dim3(nw, (n + 3) / 4), dim3(kT), lds_adj, s, d_x, d_prev, n, d, d_bits, nw);
    hipLaunchKernelGGL(fr_components, dim3(1), dim3(1024), 0, s, d_bits, n, nw, d_label, d_nc);
    if (!down(ctx, &ncl, d_nc, 1) || !hip_ok(ctx, hipStreamSynchronize(s), "sync")) return DH_ERR_HIP;
    if (ncl > 1) {
      double* d_mean = d_y;  // the whitened copy is no longer needed
      hipLaunchKernelGGL(fr_cluster_mean, dim3(n), dim3(64), 0, s, d_x, d_label, n, d, d_mean);
      hipLaunchKernelGGL(fr_recentre, dim3(gb), dim3(kT), 0, s, d_x, d_label, d_mean, n, d, d_mv);
      d_src = d_mv;
    }
  }
  hipLaunchKernelGGL(fr_colmean, dim3(d), dim3(kT), 0, s, d_src, n, d, d_m);
  hipLaunchKernelGGL(fr_cov, dim3(d * d), dim3(kT), 0, s, d_src, d_m, n, d, d_cov);
  const size_t lds_shape = ((size_t)3 * d * (d | 1) + d + 128) * 8 + (128 + d + 8) * 4;
  hipLaunchKernelGGL(fr_shape, dim3(1), dim3(64), lds_shape, s, d_cov, d, d_am, d_ax, d_ai, d_info, d_st);
  // whitened points and the radius
  hipLaunchKernelGGL(fr_matmul, dim3(gb), dim3(kT), dd * 8, s, d_x, d_ai, n, d, d_y);
  hipLaunchKernelGGL(fr_nn, dim3((n + kT - 1) / kT, reps), dim3(kT), (size_t)64 * d * 8, s, d_y, n, d, kind, d_mask,
                     d_nn);
  hipLaunchKernelGGL(fr_max, dim3(1), dim3(kT), 0, s, d_nn, (long long)reps * n, d_r);
  if ((rc = launch_ok(ctx, "friends_update launch"))) return rc;
  int st = 0;
  double info[2] = {0, 0}, r = 0.0;
  if (!down(ctx, cov, d_cov, dd) || !down(ctx, am, d_am, dd) || !down(ctx, axes, d_ax, dd) ||
      !down(ctx, axes_inv, d_ai, dd) || !down(ctx, info, d_info, 2) || !down(ctx, &st, d_st, 1) ||
      !down(ctx, &r, d_r, 1) || !hip_ok(ctx, hipStreamSynchronize(s), "sync"))
    return DH_ERR_HIP;
  if (st != DH_OK) return fail(ctx, DH_ERR_VALUE, "friends_update: covariance is not finite");
  if (info[1] > 0.5 || !(r > 0.0) || !isfinite(r))
    return fail(ctx, DH_ERR_VALUE, "friends_update: singular covariance (%d null directions) or zero radius %g",
                (int)info[1], r);
  // re-scale by the radius (bounding.py:944-953) and the volume of one shape
  const double r2 = r * r;
  for (size_t e = 0; e < dd; ++e) {
    cov[e] *= r2;
    am[e] /= r2;
    axes[e] *= r;
    axes_inv[e] /= r;
  }
  const double pref = kind == KIND_BALLS ? d * log(2.0) + d * lgamma(1.5) - lgamma(d / 2.0 + 1.0) : d * log(2.0);
  // -0.5 ln det am = 0.5 (sum log lambda + 2 d ln r)
  *logvol = pref + 0.5 * info[0] + d * log(r);
  *rmax = r;
  if (nclusters) *nclusters = ncl;
  return DH_OK;
}

int dh_friends_within(dh_ctx* ctx, const double* ctrs, int n, int d, int kind, const double* axes_inv,
                      const double* x, int m, int32_t* counts, uint64_t* bits) {
  DH_CHECK_CTX(ctx);
  if (!ctrs || n < 1 || d < 1 || d > 64 || !axes_inv || !x || m < 1 || !counts)
    return fail(ctx, DH_ERR_ARG, "friends_within: bad arguments");
  (void)hipSetDevice(ctx->device);
  arena_reset(ctx);
  const size_t nd = (size_t)n * d, md = (size_t)m * d, dd = (size_t)d * d;
  const int nw = (n + 63) / 64;
  int rc = arena_reserve(ctx, (nd + md) * 16 + dd * 8 + (size_t)m * 4 + (bits ? (size_t)m * nw * 8 : 0) + 8192);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  double* d_c = arena_up(ctx, ctrs, nd);
  double* d_x = arena_up(ctx, x, md);
  double* d_ai = arena_up(ctx, axes_inv, dd);
  double* d_ct = (double*)arena_get(ctx, nd * 8);
  double* d_xt = (double*)arena_get(ctx, md * 8);
  int* d_cnt = (int*)arena_get(ctx, (size_t)m * 4);
  unsigned long long* d_b = bits ? (unsigned long long*)arena_get(ctx, (size_t)m * nw * 8) : nullptr;
  if (!d_c || !d_x || !d_ai || !d_ct || !d_xt || !d_cnt || (bits && !d_b)) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(fr_matmul, dim3((int)((nd + kT - 1) / kT)), dim3(kT), dd * 8, s, d_c, d_ai, n, d, d_ct);
  hipLaunchKernelGGL(fr_matmul, dim3((int)((md + kT - 1) / kT)), dim3(kT), dd * 8, s, d_x, d_ai, m, d, d_xt);
  hipLaunchKernelGGL(fr_within, dim3((m + 63) / 64), dim3(64), (size_t)64 * d * 8, s, d_ct, d_xt, n, m, d, kind, d_cnt,
                     d_b, nw);
  if ((rc = launch_ok(ctx, "friends_within launch"))) return rc;
  if (!down(ctx, counts, d_cnt, (size_t)m) ||
      (bits && !down(ctx, (unsigned long long*)bits, d_b, (size_t)m * nw)))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_friends_draw(dh_ctx* ctx, const uint64_t* state6, int nsamp, const double* ctrs, int n, int d, int kind,
                    const double* axes, const double* axes_inv, int return_q, double* xs, int32_t* qs,
                    uint64_t* state6_out) {
  DH_CHECK_CTX(ctx);
  if (!state6 || nsamp < 1 || !ctrs || n < 1 || d < 1 || d > 64 || !axes || !axes_inv || !xs || !qs || !state6_out)
    return fail(ctx, DH_ERR_ARG, "friends_draw: bad arguments");
  (void)hipSetDevice(ctx->device);
  arena_reset(ctx);
  const size_t nd = (size_t)n * d, dd = (size_t)d * d;
  int rc = arena_reserve(ctx, nd * 16 + dd * 16 + (size_t)nsamp * (d * 8 + 4) + 8192);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  uint64_t* d_s = arena_up(ctx, state6, 6);
  double* d_c = arena_up(ctx, ctrs, nd);
  double* d_ax = arena_up(ctx, axes, dd);
  double* d_ai = arena_up(ctx, axes_inv, dd);
  double* d_ct = (double*)arena_get(ctx, nd * 8);
  double* d_x = (double*)arena_get(ctx, (size_t)nsamp * d * 8);
  int32_t* d_q = (int32_t*)arena_get(ctx, (size_t)nsamp * 4);
  uint64_t* d_o = (uint64_t*)arena_get(ctx, 48);
  if (!d_s || !d_c || !d_ax || !d_ai || !d_ct || !d_x || !d_q || !d_o) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(fr_matmul, dim3((int)((nd + kT - 1) / kT)), dim3(kT), dd * 8, s, d_c, d_ai, n, d, d_ct);
  hipLaunchKernelGGL(fr_draw, dim3(1), dim3(64), (size_t)3 * d * 8, s, d_s, nsamp, n, d, kind, d_c, d_ct, d_ax, d_ai,
                     return_q, d_x, d_q, d_o, ctx->zki(), ctx->zwi(), ctx->zfi());
  if ((rc = launch_ok(ctx, "friends_draw launch"))) return rc;
  if (!down(ctx, xs, d_x, (size_t)nsamp * d) || !down(ctx, qs, d_q, (size_t)nsamp) || !down(ctx, state6_out, d_o, 6))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"
