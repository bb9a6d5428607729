// Proposal kernels: one walker per lane, walker state in registers.
//
// Mapping (DESIGN.md "proposal kernels"): a walker's whole chain -- `walks`
// steps of {D normals + 1 uniform, D x D frame mat-vec, wrap/reflect, cube
// check, prior transform, log-likelihood, accept} -- is sequential, and walkers
// are independent, so each lane owns one walker and keeps u / dr / u' in
// VGPRs (kernels are instantiated per padded dimension so every array index
// is static).  The proposal frame (`axes`) and the likelihood parameters are
// wave-uniform and are fetched through the scalar cache into SGPRs; the only
// LDS traffic is the 6 KiB ziggurat table.  HBM traffic is one read of u0 and
// one write of (u, v, logl) per walker per launch -- the kernel is fp64-VALU
// bound, not bandwidth bound.
#include <stdlib.h>

#include <hiprand/hiprand_kernel.h>

#include "ctx.h"
#include "rng_pcg64.h"

using namespace dh;
#ifndef DH_RW_OCC
#define DH_RW_OCC 2
#endif

namespace {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
// DH_ABLATE is a diagnostic of the `make ablate` build only (libdynhip_ablate.so): the production kernel
// carries no ablation branches
#ifdef DH_RW_ABLATE
#define DH_ABL(a, bit) ((a).ablate & (bit))
#else
#define DH_ABL(a, bit) 0
#endif

// experiment hooks: -DDH_FENCES=<bitmask> places further never-taken branches (scheduling-region ends)
//   1 after the normals, 2 after the frame product, 4 after the cube check, 8 after the prior transform
#ifndef DH_PCG_SB
#define DH_PCG_SB 4     // scheduling barrier every 4 rows of the likelihood in the PCG64 kernel (problem.h)
#endif
#ifndef DH_PHILOX_SB
#define DH_PHILOX_SB 0  // none in the Philox kernel
#endif
#ifndef DH_FENCES
#define DH_FENCES 8  // measured best (tools/rw_exp.sh / rw_exp_run.sh): fence after the prior transform
#endif
#define DH_SCHED_FENCE(a, bit)                                \
  do {                                                        \
    if ((DH_FENCES & (bit)) && (a).fence) asm volatile("s_sleep 1"); \
  } while (0)

// random-number policy of the walk kernels
//   RNG_PCG64   numpy.random.Generator(PCG64) streams, bit for bit (ziggurat normals): the parity mode
//   RNG_PHILOX  hiprand's Philox4x32-10 device generator, keyed (seed, subsequence = seq0 + walker,
//               offset): counter based, so no generator state travels through HBM; normals are hiprand's
//               fp32 Box-Muller pairs (hiprand_normal4) widened to fp64, uniforms hiprand_uniform_double.
//               Same algorithm, same distributions to fp32 resolution of the step direction -- the
//               proposal stays exactly symmetric -- but NOT the reference's streams: the throughput mode.
enum : int { RNG_PCG64 = 0, RNG_PHILOX = 1 };

struct RwalkArgs {
  ProblemDev prob;
  int k, ndim, ncdim, walks, m;
  double scale, loglstar;
  const double* u0;
  const double* axes_t;  // m x N x N, transposed + zero padded (prep_axes_kernel)
  const int32_t* axes_idx;
  const int8_t* bc;
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
  // ensemble form (ns.hip): walker w belongs to run w / wpr; per-run threshold and
  // scale; walkers of runs whose mode differs from my_mode do nothing
  const double* run_loglstar;
  const double* run_scale;
  const int* run_mode;
  int wpr, my_mode;
  // lock-step form for host-evaluated likelihoods: do ONE proposal, write it to u,
  // its in-cube flag to nacc, the advanced stream to rng_out, and stop
  int propose_only;
  int fence;   // always 0 (see the likelihood call in rwalk_kernel)
  int ablate;  // `make ablate` build only (env DH_ABLATE): 1 no normals, 2 no frame mat-vec, 4 no likelihood, 8 no pow
  // RNG_PHILOX
  unsigned long long ph_seed, ph_seq0, ph_offset;
};

// nc N(0,1) draws of the Philox stream into the per-lane LDS column, returning their sum of squares
__device__ __forceinline__ double normals_to_lds_philox(hiprandStatePhilox4_32_10_t* st, double* dst, int lane,
                                                        int nc) {
  double ss = 0.0;
#pragma unroll 1
  for (int i = 0; i < nc; i += 4) {
    const float4 z = hiprand_normal4(st);
    const double z0 = (double)z.x, z1 = (double)z.y, z2 = (double)z.z, z3 = (double)z.w;
    dst[i * 64 + lane] = z0;
    ss = fma(z0, z0, ss);
    if (i + 1 < nc) {
      dst[(i + 1) * 64 + lane] = z1;
      ss = fma(z1, z1, ss);
    }
    if (i + 2 < nc) {
      dst[(i + 2) * 64 + lane] = z2;
      ss = fma(z2, z2, ss);
    }
    if (i + 3 < nc) {
      dst[(i + 3) * 64 + lane] = z3;
      ss = fma(z3, z3, ss);
    }
  }
  return ss;
}

// frames arrive row-major with column i = axis i (bounding.py:225-229); the walk
// kernel wants, for a fixed input index j, the N outputs contiguous so one
// s_load_dwordx16 feeds 8 FMAs: AT[f][j][i] = axes[f][i][j], zero padded to N.
__global__ void prep_axes_kernel(const double* __restrict__ axes, int m, int nc, int N,
                                 double* __restrict__ at) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = m * N * N;
  if (t >= tot) return;
  const int f = t / (N * N), r = t % (N * N), j = r / N, i = r % N;
  at[t] = (i < nc && j < nc) ? axes[(size_t)f * nc * nc + i * nc + j] : 0.0;
}

// generic_random_walk (internal_samplers.py:866-986), one walker per lane.
// FULL: ndim == ncdim == N (all guards fold away).  KIND: problem.h.
template <int N, bool FULL, int KIND, int RNG>
__global__ void __launch_bounds__(64, DH_RW_OCC) rwalk_kernel(RwalkArgs a) {
  __shared__ ZigLds zig;
  __shared__ double sx[N * 64];  // per-lane column: dr, then v = prior(u')
  if constexpr (RNG == RNG_PCG64) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int lane = threadIdx.x;
  const int w = blockIdx.x * 64 + lane;
  const bool live = w < a.k;
  const int wi = live ? w : a.k - 1;  // dead lanes shadow the last walker (no stores)
  const int n = FULL ? N : a.ndim, nc = FULL ? N : a.ncdim;
  double loglstar = a.loglstar, scale = a.scale;
  if (a.run_mode) {
    const int run = wi / a.wpr;
    if (a.run_mode[run] != a.my_mode) return;
    loglstar = a.run_loglstar[run];
    scale = a.run_scale[run];
  }

  double u[N], up[N], acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) u[i] = (FULL || i < n) ? a.u0[(size_t)wi * n + i] : 0.5;
  Pcg64 g;
  hiprandStatePhilox4_32_10_t ph;
  if constexpr (RNG == RNG_PCG64)
    g.load(a.rng_in + (size_t)wi * 4);
  else
    hiprand_init(a.ph_seed, a.ph_seq0 + (unsigned long long)wi, a.ph_offset, &ph);
  const int my_frame = a.axes_idx ? a.axes_idx[wi] : 0;

  int nacc = 0, nrej = 0;
  double logl_cur = 0.0;
  const double inv_nc = 1.0 / (double)nc;

#pragma unroll 1
  for (int step = 0; step < a.walks; ++step) {
    // propose_ball_point (internal_samplers.py:989-1035): non-cluster dims are
    // redrawn first (rstate.random(n - n_cluster)) ...
    if (!FULL) {
#pragma unroll 1
      for (int i = nc; i < n; ++i)
        sx[i * 64 + lane] = RNG == RNG_PCG64 ? g.next_double() : hiprand_uniform_double(&ph);
    }
    // ... then randsphere (bounding.py:1288-1297): nc normals, one uniform
    double ss = 0.0;
    if (DH_ABL(a, 1)) {
#pragma unroll 1
      for (int i = 0; i < nc; ++i) {
        const double x = 0.1 * (i + 1);
        sx[i * 64 + lane] = x;
        ss = fma(x, x, ss);
      }
    } else if constexpr (RNG == RNG_PCG64) {
      ss = normals_to_lds(g, &zig, sx, lane, nc);
    } else {
      ss = normals_to_lds_philox(&ph, sx, lane, nc);
    }
    DH_SCHED_FENCE(a, 1);
    const double ur = RNG == RNG_PCG64 ? g.next_double() : hiprand_uniform_double(&ph);
    const double fac = scale * ((DH_ABL(a, 8) ? ur : pow(ur, inv_nc)) / sqrt(ss));
    // du = axes @ dr, frame wave-uniform: waterfall over the distinct frames
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.0;
    bool done = false;
    while (!done) {
      const int cur = __builtin_amdgcn_readfirstlane(my_frame);
      if (cur == my_frame) {
        // frame rows through the scalar cache: the pointer is pinned to SGPRs (see as_const_uniform)
        if (!DH_ABL(a, 2)) matvec_sgpr<N>(as_const_uniform(a.axes_t + (size_t)cur * N * N), sx, lane, nc, acc);
        done = true;
      }
    }
    DH_SCHED_FENCE(a, 2);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (FULL || i < nc)
        up[i] = fma(fac, acc[i], u[i]);
      else
        up[i] = (i < n) ? sx[i * 64 + lane] : 0.5;
    }
    // periodic wrap / reflection, then unitcheck (utils.py:1036-1050)
    bool inside = true;
    if (a.bc) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (FULL || i < n) {
          const int b = a.bc[i];
          double x = up[i];
          if (b == DH_BC_PERIODIC) x = wrap01(x);
          if (b == DH_BC_REFLECT) x = reflect01(x);
          up[i] = x;
          if (b == DH_BC_HARD)
            inside = inside && (x > 0.0) && (x < 1.0);
          else
            inside = inside && (x > -0.5) && (x < 1.5);
        }
      }
    } else {
      double lo = up[0], hi = up[0];
#pragma unroll
      for (int i = 1; i < N; ++i) {
        lo = fmin(lo, up[i]);
        hi = fmax(hi, up[i]);
      }
      inside = (lo > 0.0) && (hi < 1.0);  // padded entries sit at 0.5
    }
    if (a.propose_only) {
      if (live) {
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (FULL || i < n) a.u[(size_t)w * n + i] = up[i];
        a.nacc[w] = inside ? 1 : 0;
        if (RNG == RNG_PCG64 && a.rng_out) g.store(a.rng_out + (size_t)w * 4);
      }
      return;
    }
    if (!inside) {  // counted as a call and a reject, no likelihood evaluated
      ++nrej;
      continue;
    }
    DH_SCHED_FENCE(a, 4);
    prior_to_lds<N, FULL, KIND>(a.prob, up, n, sx, lane);
    DH_SCHED_FENCE(a, 8);
    // `a.fence` is always 0, but the compiler cannot know: the never-taken branch ends the scheduling
    // region in front of the 325 unrolled FMAs of the likelihood.  As one region with the frame product
    // the scheduler clusters their scalar loads and spills 690 SGPRs (kernel +20 %); measured, not guessed:
    // tools/rw_ablate.sh, -Rpass-analysis=kernel-resource-usage.
    double ll;
    if (a.fence | DH_ABL(a, 4))
      ll = loglstar + ur - 0.6;
    else
      ll = loglike_lds<N, FULL, KIND, (RNG == RNG_PCG64 ? DH_PCG_SB : DH_PHILOX_SB)>(a.prob, n, sx, lane, acc);
    if (ll > loglstar) {
#pragma unroll
      for (int i = 0; i < N; ++i) u[i] = up[i];
      logl_cur = ll;
      ++nacc;
    } else {
      ++nrej;
    }
  }
  // v of the returned point; logl is re-evaluated when nothing was accepted
  // (internal_samplers.py:970-975)
  prior_to_lds<N, FULL, KIND>(a.prob, u, n, sx, lane);
  if (nacc == 0) logl_cur = loglike_lds<N, FULL, KIND>(a.prob, n, sx, lane, acc);
  if (live) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (FULL || i < n) {
        a.u[(size_t)w * n + i] = u[i];
        a.v[(size_t)w * n + i] = sx[i * 64 + lane];
      }
    a.logl[w] = logl_cur;
    a.nacc[w] = nacc;
    a.nrej[w] = nrej;
    if (RNG == RNG_PCG64 && a.rng_out) g.store(a.rng_out + (size_t)w * 4);
  }
}

// ---------------------------------------------------------------------------
template <int N>
__global__ void __launch_bounds__(64)
    eval_kernel(ProblemDev prob, int k, const double* __restrict__ u, double* v, double* logl) {
  __shared__ double sx[N * 64];
  const int lane = threadIdx.x;
  const int w = blockIdx.x * 64 + lane;
  const int wi = w < k ? w : k - 1;
  const int n = prob.ndim;
  double uu[N], acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) uu[i] = (i < n) ? u[(size_t)wi * n + i] : 0.5;
  prior_to_lds<N, false, KIND_GENERIC>(prob, uu, n, sx, lane);
  const double ll = loglike_lds<N, false, KIND_GENERIC>(prob, n, sx, lane, acc);
  if (w >= k) return;
  logl[w] = ll;
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) v[(size_t)w * n + i] = sx[i * 64 + lane];
}

__global__ void seed_kernel(const uint32_t* __restrict__ entropy, int nwords, uint32_t first, int k,
                            uint64_t* states) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= k) return;
  Pcg64 g;
  seed_from_child(g, entropy, nwords, first + (uint32_t)w);
  g.store(states + (size_t)w * 4);
}

__global__ void __launch_bounds__(64)
    stream_kernel(const uint64_t* state_in, int n_normal, int n_unif, double* normals, double* unifs,
                  uint64_t* state_out, const uint64_t* zki, const uint64_t* zwi, const uint64_t* zfi) {
  __shared__ ZigLds zig;
  zig_stage(&zig, zki, zwi, zfi);
  if (threadIdx.x != 0) return;
  Pcg64 g;
  g.load(state_in);
  for (int i = 0; i < n_normal; ++i) normals[i] = std_normal(g, &zig);
  for (int i = 0; i < n_unif; ++i) unifs[i] = g.next_double();
  g.store(state_out);
}


}  // namespace

extern "C" {

int dh_set_rwalk_form(dh_ctx* ctx, int form) {
  DH_CHECK_CTX(ctx);
  if (form < 0 || form > 2)
    return fail(ctx, DH_ERR_ARG, "rwalk form %d (0 / 2 = four lanes per walker where built, 1 = lane per walker)", form);
  ctx->rwalk_form = form;
  return DH_OK;
}

int dh_set_rwalk_items(dh_ctx* ctx, int on, long long budget_bytes) {
  DH_CHECK_CTX(ctx);
  if (budget_bytes < 0) return fail(ctx, DH_ERR_ARG, "rwalk item-stream budget %lld", budget_bytes);
  ctx->rwalk_items = on ? 1 : 0;
  if (budget_bytes > 0) ctx->items_budget = (size_t)budget_bytes;
  return DH_OK;
}

int dh_rwalk_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                       const double* axes, int m, const int32_t* axes_idx, double scale,
                       double loglstar, int walks, const int8_t* bc, const uint64_t* rng, double* u,
                       double* v, double* logl, int32_t* naccept, int32_t* nreject,
                       uint64_t* rng_out) {
  return dh::rwalk_launch_runs(ctx, problem, k, ndim, ncdim, u0, axes, m, axes_idx, scale, loglstar, walks,
                               bc, rng, u, v, logl, naccept, nreject, rng_out, nullptr, nullptr, nullptr, 1,
                               0, nullptr);
}

int dh_rwalk_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                              const double* axes, int m, const int32_t* axes_idx, double scale,
                              double loglstar, int walks, const int8_t* bc, uint64_t seed, uint64_t sequence0,
                              uint64_t offset, double* u, double* v, double* logl, int32_t* naccept,
                              int32_t* nreject) {
  const dh::PhiloxKey key = {seed, sequence0, offset};
  return dh::rwalk_launch_runs(ctx, problem, k, ndim, ncdim, u0, axes, m, axes_idx, scale, loglstar, walks,
                               bc, nullptr, u, v, logl, naccept, nreject, nullptr, nullptr, nullptr, nullptr,
                               1, 0, &key);
}

}  // extern "C"

int dh::rwalk_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                          const double* axes, int m, const int32_t* axes_idx, double scale,
                          double loglstar, int walks, const int8_t* bc, const uint64_t* rng, double* u,
                          double* v, double* logl, int32_t* naccept, int32_t* nreject, uint64_t* rng_out,
                          const double* run_loglstar, const double* run_scale, const int* run_mode,
                          int wpr, int my_mode, const dh::PhiloxKey* philox) {
  DH_CHECK_CTX(ctx);
  RwalkArgs a;
  a.ph_seed = philox ? philox->seed : 0;
  a.ph_seq0 = philox ? philox->seq0 : 0;
  a.ph_offset = philox ? philox->offset : 0;
  if (!philox && !rng && k > 0) return fail(ctx, DH_ERR_ARG, "rwalk: no generator states");
  a.run_loglstar = run_loglstar;
  a.run_scale = run_scale;
  a.run_mode = run_mode;
  a.wpr = wpr;
  a.my_mode = my_mode;
  a.propose_only = problem == -1 ? 1 : 0;
  if (a.propose_only) {
    a.prob = ProblemDev();
    a.prob.ndim = ndim;
    a.prob.like_id = 99;   // never evaluated
    a.prob.prior_id = 99;
  } else {
    if (!get_problem(ctx, problem, &a.prob)) return DH_ERR_ARG;
    if (a.prob.ndim != ndim) return fail(ctx, DH_ERR_ARG, "problem ndim %d != %d", a.prob.ndim, ndim);
  }
  if (k <= 0) return DH_OK;
  if (ncdim < 1 || ncdim > ndim || m < 1 || walks < 1)
    return fail(ctx, DH_ERR_ARG, "rwalk: ncdim=%d ndim=%d m=%d walks=%d", ncdim, ndim, m, walks);
  if (ndim > kMaxRegDim)
    return wide_walk_launch(ctx, 0, problem, k, ndim, ncdim, u0, axes, m, axes_idx, scale, loglstar, walks,
                            0, bc, rng, u, v, logl, naccept, nreject, nullptr, nullptr, rng_out, run_loglstar,
                            run_scale, run_mode, nullptr, wpr, my_mode, philox);
  // Four lanes per walker + matrix cores (walkq.hip): built for full-dimensional proposals,
  // 2 <= ndim <= 32, fused likelihood and prior, any boundary conditions.  It does the same
  // walk on the same streams (counts and generator states identical, coordinates to rounding).  Round 4: with
  // the PCG64 streams written out by a generator pass of their own it is the faster form at every launch size
  // (64 x 512 walkers: 0.27 ms against 0.86; 64 x 2000: 1.02 against 1.23), so form 0 takes it whenever it
  // applies -- a function of the problem alone, never of the launch size or the device, so that a run's
  // accept / reject sequence cannot depend on how many runs share a GPU (form 1: never, 2: same as 0).
  const bool quad_ok = !a.propose_only && ndim == ncdim && ndim >= 2 && ndim <= kMaxRegDim &&
                       (long long)walks * (ndim + 1) < (1ll << 24);
  if (quad_ok && ctx->rwalk_form != 1)
    return rwalkq_launch(ctx, a.prob, k, ndim, u0, axes, m, axes_idx, scale, loglstar, walks, rng, u, v, logl,
                         naccept, nreject, rng_out, run_loglstar, run_scale, run_mode, wpr, my_mode, philox, bc);
  a.k = k;
  a.ndim = ndim;
  a.ncdim = ncdim;
  a.walks = walks;
  a.m = m;
  a.scale = scale;
  a.loglstar = loglstar;
  a.u0 = u0;
  a.axes_idx = axes_idx;
  a.bc = bc;
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
  a.ablate = 0;
  a.fence = 0;
#ifdef DH_RW_ABLATE
  {
    const char* e = getenv("DH_ABLATE");
    a.ablate = e ? atoi(e) : 0;
  }
#endif
  const int N = pad_dim(ndim);
  // transposed + padded copy of the frames (stream ordered, context scratch)
  const size_t at_bytes = (size_t)m * N * N * sizeof(double);
  if (at_bytes > ctx->axes_t_cap) {
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    if (ctx->axes_t) (void)hipFree(ctx->axes_t);
    ctx->axes_t = nullptr;
    ctx->axes_t_cap = 0;
    if (!hip_ok(ctx, hipMalloc((void**)&ctx->axes_t, at_bytes * 2), "hipMalloc(axes_t)"))
      return DH_ERR_NOMEM;
    ctx->axes_t_cap = at_bytes * 2;
  }
  hipLaunchKernelGGL(prep_axes_kernel, dim3((m * N * N + 255) / 256), dim3(256), 0, ctx->stream, axes, m,
                     ncdim, N, ctx->axes_t);
  a.axes_t = ctx->axes_t;
  const dim3 grid((k + 63) / 64), block(64);
  const bool full = (ndim == N && ncdim == N) && !a.propose_only;
  const int kind = full ? problem_kind(a.prob.like_id, a.prob.prior_id) : KIND_GENERIC;
#define L(NN, FF, KK)                                                                            \
  do {                                                                                           \
    if (philox)                                                                                  \
      hipLaunchKernelGGL((rwalk_kernel<NN, FF, KK, RNG_PHILOX>), grid, block, 0, ctx->stream, a); \
    else                                                                                         \
      hipLaunchKernelGGL((rwalk_kernel<NN, FF, KK, RNG_PCG64>), grid, block, 0, ctx->stream, a);  \
  } while (0)
#define X(NN)                                       \
  if (N == NN) {                                    \
    if (!full)                                      \
      L(NN, false, KIND_GENERIC);                   \
    else if (kind == KIND_PREC_AFFINE)              \
      L(NN, true, KIND_PREC_AFFINE);                \
    else if (kind == KIND_IID_AFFINE)               \
      L(NN, true, KIND_IID_AFFINE);                 \
    else if (kind == KIND_EGGBOX_IDENTITY)          \
      L(NN, true, KIND_EGGBOX_IDENTITY);            \
    else if (kind == KIND_IID_NORMAL)               \
      L(NN, true, KIND_IID_NORMAL);                 \
    else                                            \
      L(NN, true, KIND_GENERIC);                    \
  }
  DH_DIM_LIST(X)
#undef X
#undef L
  return hip_ok(ctx, hipGetLastError(), "rwalk launch") ? DH_OK : DH_ERR_HIP;
}

extern "C" {


int dh_rwalk_propose(dh_ctx* ctx, int k, int ndim, int ncdim, const double* u0, const double* axes, int m,
                     const int32_t* axes_idx, double scale, const int8_t* bc, const uint64_t* rng,
                     double* u_prop, int32_t* inside, uint64_t* rng_out) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!u0 || !axes || !rng || !u_prop || !inside || !rng_out)
    return fail(ctx, DH_ERR_ARG, "rwalk_propose: null pointer");
  if (ndim > kMaxRegDim)
    return fail(ctx, DH_ERR_ARG, "rwalk_propose: ndim=%d > %d not built", ndim, kMaxRegDim);
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  int rc = arena_reserve(ctx, 2 * kd * 8 + (size_t)m * ncdim * ncdim * 8 + (size_t)k * (4 + 4 + 64) +
                                  (size_t)ndim + 8192);
  if (rc) return rc;
  const double* d_u0 = arena_up(ctx, u0, kd);
  const double* d_axes = arena_up(ctx, axes, (size_t)m * ncdim * ncdim);
  const int32_t* d_idx = axes_idx ? arena_up(ctx, axes_idx, (size_t)k) : nullptr;
  const int8_t* d_bc = bc ? arena_up(ctx, bc, (size_t)ndim) : nullptr;
  const uint64_t* d_rng = arena_up(ctx, rng, (size_t)k * 4);
  double* d_u = (double*)arena_get(ctx, kd * 8);
  int32_t* d_in = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_ro = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  if (!d_u0 || !d_axes || !d_rng || !d_u || !d_in || !d_ro || (axes_idx && !d_idx) || (bc && !d_bc))
    return DH_ERR_NOMEM;
  rc = dh::rwalk_launch_runs(ctx, -1, k, ndim, ncdim, d_u0, d_axes, m, d_idx, scale, 0.0, 1, d_bc, d_rng, d_u,
                             nullptr, nullptr, d_in, nullptr, d_ro, nullptr, nullptr, nullptr, 1, 0, nullptr);
  if (rc) return rc;
  if (!down(ctx, u_prop, d_u, kd) || !down(ctx, inside, d_in, (size_t)k) ||
      !down(ctx, rng_out, d_ro, (size_t)k * 4))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_rwalk_batch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                   const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                   int walks, const int8_t* bc, const uint64_t* rng, double* u, double* v,
                   double* logl, int32_t* naccept, int32_t* nreject, uint64_t* rng_out) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!u0 || !axes || !rng || !u || !v || !logl || !naccept || !nreject)
    return fail(ctx, DH_ERR_ARG, "rwalk: null pointer");
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  size_t need = 3 * kd * 8 + (size_t)m * ncdim * ncdim * 8 + (size_t)k * (4 + 8 + 4 + 4 + 64) +
                (size_t)ndim + 16 * 256;
  int rc = arena_reserve(ctx, need);
  if (rc) return rc;
  const double* d_u0 = arena_up(ctx, u0, kd);
  const double* d_axes = arena_up(ctx, axes, (size_t)m * ncdim * ncdim);
  const int32_t* d_idx = axes_idx ? arena_up(ctx, axes_idx, (size_t)k) : nullptr;
  const int8_t* d_bc = bc ? arena_up(ctx, bc, (size_t)ndim) : nullptr;
  const uint64_t* d_rng = arena_up(ctx, rng, (size_t)k * 4);
  double* d_u = (double*)arena_get(ctx, kd * 8);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_logl = (double*)arena_get(ctx, (size_t)k * 8);
  int32_t* d_na = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_nr = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_ro = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  if (!d_u0 || !d_axes || !d_rng || !d_u || !d_v || !d_logl || !d_na || !d_nr || !d_ro ||
      (axes_idx && !d_idx) || (bc && !d_bc))
    return DH_ERR_NOMEM;
  rc = dh_rwalk_batch_dev(ctx, problem, k, ndim, ncdim, d_u0, d_axes, m, d_idx, scale, loglstar, walks,
                          d_bc, d_rng, d_u, d_v, d_logl, d_na, d_nr, d_ro);
  if (rc) return rc;
  if (!down(ctx, u, d_u, kd) || !down(ctx, v, d_v, kd) || !down(ctx, logl, d_logl, (size_t)k) ||
      !down(ctx, naccept, d_na, (size_t)k) || !down(ctx, nreject, d_nr, (size_t)k) ||
      !down(ctx, rng_out, d_ro, (size_t)k * 4))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_rwalk_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                          const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                          int walks, const int8_t* bc, uint64_t seed, uint64_t sequence0, uint64_t offset,
                          double* u, double* v, double* logl, int32_t* naccept, int32_t* nreject) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!u0 || !axes || !u || !v || !logl || !naccept || !nreject)
    return fail(ctx, DH_ERR_ARG, "rwalk (philox): null pointer");
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  size_t need = 3 * kd * 8 + (size_t)m * ncdim * ncdim * 8 + (size_t)k * (4 + 8 + 4 + 4) + (size_t)ndim + 16 * 256;
  int rc = arena_reserve(ctx, need);
  if (rc) return rc;
  const double* d_u0 = arena_up(ctx, u0, kd);
  const double* d_axes = arena_up(ctx, axes, (size_t)m * ncdim * ncdim);
  const int32_t* d_idx = axes_idx ? arena_up(ctx, axes_idx, (size_t)k) : nullptr;
  const int8_t* d_bc = bc ? arena_up(ctx, bc, (size_t)ndim) : nullptr;
  double* d_u = (double*)arena_get(ctx, kd * 8);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_logl = (double*)arena_get(ctx, (size_t)k * 8);
  int32_t* d_na = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_nr = (int32_t*)arena_get(ctx, (size_t)k * 4);
  if (!d_u0 || !d_axes || !d_u || !d_v || !d_logl || !d_na || !d_nr || (axes_idx && !d_idx) || (bc && !d_bc))
    return DH_ERR_NOMEM;
  rc = dh_rwalk_batch_philox_dev(ctx, problem, k, ndim, ncdim, d_u0, d_axes, m, d_idx, scale, loglstar, walks,
                                 d_bc, seed, sequence0, offset, d_u, d_v, d_logl, d_na, d_nr);
  if (rc) return rc;
  if (!down(ctx, u, d_u, kd) || !down(ctx, v, d_v, kd) || !down(ctx, logl, d_logl, (size_t)k) ||
      !down(ctx, naccept, d_na, (size_t)k) || !down(ctx, nreject, d_nr, (size_t)k))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"

int dh::eval_launch_dev(dh_ctx* ctx, int problem, int k, const double* u, double* v, double* logl) {
  ProblemDev p;
  if (!get_problem(ctx, problem, &p)) return DH_ERR_ARG;
  if (k <= 0) return DH_OK;
  if (p.ndim > kMaxRegDim) return wide_eval_launch(ctx, p, k, u, v, logl);
  const dim3 grid((k + 63) / 64), block(64);
  const int ndim = p.ndim;
  bool hit = false;
#define X(NN)                                                                  \
  if (!hit && ndim <= NN) {                                                    \
    hit = true;                                                                \
    hipLaunchKernelGGL(eval_kernel<NN>, grid, block, 0, ctx->stream, p, k, u, v, logl); \
  }
  DH_DIM_LIST(X)
#undef X
  return hip_ok(ctx, hipGetLastError(), "eval launch") ? DH_OK : DH_ERR_HIP;
}

extern "C" {

int dh_problem_eval(dh_ctx* ctx, int problem, int k, const double* u, double* v, double* logl) {
  DH_CHECK_CTX(ctx);
  ProblemDev p;
  if (!get_problem(ctx, problem, &p)) return DH_ERR_ARG;
  if (k <= 0) return DH_OK;
  const int ndim = p.ndim;
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  int rc = arena_reserve(ctx, 2 * kd * 8 + (size_t)k * 8 + 4096);
  if (rc) return rc;
  const double* d_u = arena_up(ctx, u, kd);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_l = (double*)arena_get(ctx, (size_t)k * 8);
  if (!d_u || !d_v || !d_l) return DH_ERR_NOMEM;
  const dim3 grid((k + 63) / 64), block(64);
  bool hit = false;
  if (ndim > kMaxRegDim) {
    hit = true;
    rc = wide_eval_launch(ctx, p, k, d_u, d_v, d_l);
    if (rc) return rc;
  }
#define X(NN)                                                                      \
  if (!hit && ndim <= NN) {                                                        \
    hit = true;                                                                    \
    hipLaunchKernelGGL(eval_kernel<NN>, grid, block, 0, ctx->stream, p, k, d_u, d_v, d_l); \
  }
  DH_DIM_LIST(X)
#undef X
  if (!hip_ok(ctx, hipGetLastError(), "eval launch")) return DH_ERR_HIP;
  if (!down(ctx, v, d_v, kd) || !down(ctx, logl, d_l, (size_t)k)) return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_seed_children(dh_ctx* ctx, const uint32_t* entropy_words, int n_words, uint32_t first_child,
                     int k, uint64_t* states) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!entropy_words || n_words < 1 || n_words > 64 || !states)
    return fail(ctx, DH_ERR_ARG, "seed_children: n_words=%d", n_words);
  arena_reset(ctx);
  int rc = arena_reserve(ctx, (size_t)k * 32 + 1024);
  if (rc) return rc;
  const uint32_t* d_e = arena_up(ctx, entropy_words, (size_t)n_words);
  uint64_t* d_s = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  if (!d_e || !d_s) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(seed_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, d_e, n_words,
                     first_child, k, d_s);
  if (!hip_ok(ctx, hipGetLastError(), "seed launch")) return DH_ERR_HIP;
  if (!down(ctx, states, d_s, (size_t)k * 4)) return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_rng_stream(dh_ctx* ctx, const uint64_t* state4, int n_normal, int n_unif, double* normals,
                  double* unifs, uint64_t* state4_out) {
  DH_CHECK_CTX(ctx);
  if (!state4 || n_normal < 0 || n_unif < 0) return fail(ctx, DH_ERR_ARG, "rng_stream: bad args");
  arena_reset(ctx);
  int rc = arena_reserve(ctx, (size_t)(n_normal + n_unif) * 8 + 4096);
  if (rc) return rc;
  const uint64_t* d_s = arena_up(ctx, state4, 4);
  double* d_n = (double*)arena_get(ctx, (size_t)(n_normal + 1) * 8);
  double* d_u = (double*)arena_get(ctx, (size_t)(n_unif + 1) * 8);
  uint64_t* d_o = (uint64_t*)arena_get(ctx, 32);
  if (!d_s || !d_n || !d_u || !d_o) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(stream_kernel, dim3(1), dim3(64), 0, ctx->stream, d_s, n_normal, n_unif, d_n, d_u,
                     d_o, ctx->zki(), ctx->zwi(), ctx->zfi());
  if (!hip_ok(ctx, hipGetLastError(), "stream launch")) return DH_ERR_HIP;
  if (!down(ctx, normals, d_n, (size_t)n_normal) || !down(ctx, unifs, d_u, (size_t)n_unif) ||
      !down(ctx, state4_out, d_o, 4))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"
