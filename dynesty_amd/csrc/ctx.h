// Internal: the context object behind dh_ctx* and small launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/dynhip.h"
#include "problem.h"

// Walker / candidate state lives in registers, so the per-lane kernels are
// instantiated for a fixed list of padded dimensions.
#ifndef DH_DIM_LIST  // (a reduced list speeds up compile-time experiments: -D"DH_DIM_LIST(X)=X(25)")
#define DH_DIM_LIST(X) X(1) X(2) X(3) X(4) X(5) X(6) X(8) X(10) X(12) X(16) X(20) X(25) X(32)
#endif
constexpr int kMaxRegDim = 32;
// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: the memo of what was already
// set is kept per device ordinal (a process may hold contexts on several GPUs)
constexpr int kMaxDev = 64;
#define DH_DEV_MEMO(name) static size_t name##_dev[kMaxDev] = {}; size_t& name = name##_dev[ctx->device & (kMaxDev - 1)]
inline int pad_dim(int n) {
#define X(NN) \
  if (n <= NN) return NN;
  DH_DIM_LIST(X)
#undef X
  return 0;
}

struct dh_problem_rec {
  bool live = false;
  int ndim = 0;
  int like_id = 0, prior_id = 0;
  double* like_par = nullptr;   // device
  double* prior_par = nullptr;  // device
  double* prec_t = nullptr;     // device: padded precision matrix (GAUSS_PREC)
  int n_like = 0, n_prior = 0;
};

struct dh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<dh_problem_rec> problems;
  // ziggurat tables in device memory (bit patterns)
  uint64_t* zig = nullptr;  // [ki | wi | fi] 3*256
  // grow-only staging arena for the host-pointer entry points
  char* arena = nullptr;
  size_t arena_cap = 0;
  size_t arena_top = 0;
  // transposed/padded proposal frames for the walk kernels
  double* axes_t = nullptr;
  size_t axes_t_cap = 0;
  // scratch of the rebuild kernel (permutations, node table, per-node ellipsoids)
  char* rebuild_ws = nullptr;
  size_t rebuild_ws_cap = 0;
  unsigned rebuild_epoch = 0;  // rebuild launches so far (tags of the k-means partials, rebuild.hip)
  // side stream of the rebuild: the root's full eigen-system is solved there while the tree is built
  // on `stream` (fork after k_root, join before k_finish); created on first use
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_leaf = nullptr;
  // rwalk kernel form (dh_set_rwalk_form): 0 / 2 = four lanes per walker (walkq.hip) wherever that kernel is
  // built -- decided by the problem alone, never by the launch size --, 1 = one walker per lane always
  int rwalk_form = 0;
  // DH_COOP_LAUNCH=1: kernels whose WHOLE grid meets at spin barriers (k_root_parts, wide_eig*_kernel) go through
  // hipLaunchCooperativeKernel, so that the runtime vouches for the co-residency the library otherwise derives from
  // the occupancy query (and guards with a spin limit that fails the run instead of hanging the device).  Off by
  // default: the cooperative path costs launch latency on the rebuild's critical path (EXPERIMENTS.md).
  int coop_launch = 0;
  // unit-cube sampler form (env DH_CUBE_FORM): 0 = four lanes per walker for launches that would leave SIMDs empty with
  // one walker per lane, 1 = one walker per lane always, 2 = four lanes always (PCG64 streams, ndim <= 32)
  int cube_form = 0;
  int num_cu = 256;  // hipDeviceProp_t::multiProcessorCount
  // options of the resident run loop (dh_ns_set_option, keys DH_NS_OPT_* of dynhip.h); NaN = the reference's default
  // per-dimension DH_BC_* flags for the resident loop's proposals (dh_ns_set_boundary); empty = all hard
  std::vector<int8_t> ns_bc;
  double ns_opt[8] = {__builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan(""),
                      __builtin_nan(""), __builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
  // rwalk, four lanes per walker: the walkers' PCG64 item streams written out by a generator pass ahead of the walk
  // (walkq.hip: itemgen_kernel; env DH_RWALK_ITEMS=0 keeps the generator inside the walk kernel).  Grow-only
  // buffer, launches larger than the budget go in chunks of walkers
  int rwalk_items = 1;
  int itemgen_blocks_per_cu = 0;  // occupancy of itemgen_kernel on this context's device, asked once (walkq.hip)
  // A request of the resident loop to the NEXT generator pass (round 6): `runs` extra workgroups in front of
  // itemgen_kernel's grid sort each run's n live log-likelihoods by (value, slot) into out[run * stride ...] -- the
  // order ns_consume needs for the same fill, formed in the shadow of the walk instead of at the head of the consumption.
  // Cleared by the launch that serves it (done = 1); a launch that cannot (chunked, another generator) leaves done = 0.
  struct PresortReq {
    const double* keys = nullptr;  // runs x n
    unsigned short* out = nullptr;
    int n = 0, runs = 0, stride = 0, done = 0;
  } presort;
  double* items = nullptr;
  size_t items_cap = 0;
  size_t items_budget = (size_t)1 << 30;

  const uint64_t* zki() const { return zig; }
  const uint64_t* zwi() const { return zig + 256; }
  const uint64_t* zfi() const { return zig + 512; }
  const uint64_t* pcg_jump() const { return zig + 768; }  // G_1 .. G_64 as (hi, lo)
};

namespace dh {

int fail(dh_ctx* ctx, int code, const char* fmt, ...);
bool hip_ok(dh_ctx* ctx, hipError_t e, const char* what);

// bump allocator over the context arena (256-byte aligned); reset per call
void arena_reset(dh_ctx* ctx);
void* arena_get(dh_ctx* ctx, size_t bytes);  // nullptr + error set on failure
int arena_reserve(dh_ctx* ctx, size_t bytes);

template <typename T>
inline T* arena_up(dh_ctx* ctx, const T* host, size_t count) {
  T* d = (T*)arena_get(ctx, count * sizeof(T));
  if (!d) return nullptr;
  if (host && count) {
    if (!hip_ok(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream),
                "H2D"))
      return nullptr;
  }
  return d;
}

template <typename T>
inline bool down(dh_ctx* ctx, T* host, const T* dev, size_t count) {
  if (!host || !count) return true;
  return hip_ok(ctx, hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream),
                "D2H");
}

bool get_problem(dh_ctx* ctx, int handle, ProblemDev* out);

// key of the throughput-mode generator (hiprand Philox4x32-10): walker w draws from subsequence
// seq0 + w of `seed`, starting `offset` draws in
struct PhiloxKey {
  unsigned long long seed, seq0, offset;
};

// ensemble forms used by ns.hip: per-run loglstar / scale arrays, walkers of
// runs whose run_mode != my_mode are skipped (run = walker / wpr)
int rwalk_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                      const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                      int walks, const int8_t* bc, const uint64_t* rng, double* u, double* v, double* logl,
                      int32_t* naccept, int32_t* nreject, uint64_t* rng_out, const double* run_loglstar,
                      const double* run_scale, const int* run_mode, int wpr, int my_mode,
                      const PhiloxKey* philox = nullptr);
// Launch of a kernel whose whole grid must be resident at once (see dh_ctx::coop_launch).
template <class A0, class A1>
inline hipError_t launch_all_resident(dh_ctx* ctx, void (*fn)(A0, A1), dim3 grid, dim3 block, size_t lds, A0 a0, A1 a1) {
  if (ctx->coop_launch) {
    void* args[2] = {(void*)&a0, (void*)&a1};
    return hipLaunchCooperativeKernel((const void*)fn, grid, block, args, (unsigned)lds, ctx->stream);
  }
  hipLaunchKernelGGL(fn, grid, block, lds, ctx->stream, a0, a1);
  return hipSuccess;
}
template <class A0>
inline hipError_t launch_all_resident(dh_ctx* ctx, void (*fn)(A0), dim3 grid, dim3 block, size_t lds, A0 a0) {
  if (ctx->coop_launch) {
    void* args[1] = {(void*)&a0};
    return hipLaunchCooperativeKernel((const void*)fn, grid, block, args, (unsigned)lds, ctx->stream);
  }
  hipLaunchKernelGGL(fn, grid, block, lds, ctx->stream, a0);
  return hipSuccess;
}

// walkq.hip: the same walk with four lanes per walker (ndim == ncdim in 2..32)
int rwalkq_launch(dh_ctx* ctx, const ProblemDev& prob, int k, int ndim, const double* u0, const double* axes, int m,
                  const int32_t* axes_idx, double scale, double loglstar, int walks, const uint64_t* rng, double* u,
                  double* v, double* logl, int32_t* naccept, int32_t* nreject, uint64_t* rng_out,
                  const double* run_loglstar, const double* run_scale, const int* run_mode, int wpr, int my_mode,
                  const PhiloxKey* philox, const int8_t* bc = nullptr);
int slice_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                      const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                      int slices, int doubling, const uint64_t* rng, double* u, double* v, double* logl,
                      int32_t* ncalls, int32_t* nexpand, int32_t* ncontract, int32_t* flags,
                      uint64_t* rng_out, const double* run_loglstar, const double* run_scale,
                      const int* run_mode, const int* run_doubling, int wpr, int my_mode,
                      const PhiloxKey* philox = nullptr);
int unif_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                     const double* axes, const double* ams, const double* cumprob, double loglstar,
                     const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u, double* v,
                     double* logl, int32_t* ncalls, int32_t* flags, uint64_t* rng_out,
                     const double* run_loglstar, const int* run_mode, int wpr, int my_mode,
                     const PhiloxKey* philox = nullptr, const int* run_nells = nullptr, int run_me = 0);

int rebuild_launch_full(dh_ctx* ctx, int runs, const double* pts, int n, int d, int mode, int max_ells,
                        int32_t* nells, int32_t* status, double* ctrs, double* covs, double* ams,
                        double* axes, double* axlens, double* logvols, int32_t* leaf_of_point,
                        int32_t* nnodes, const int* active, const int* n_arr);
int rebuild_launch_masked(dh_ctx* ctx, int runs, const double* pts, int n, int d, int mode, int max_ells,
                          int32_t* nells, int32_t* status, double* ctrs, double* covs, double* ams,
                          double* axes, double* axlens, double* logvols, const int* active);
int enlarge_launch_masked(dh_ctx* ctx, int runs, int max_ells, const int32_t* nells, int d, double* covs,
                          double* ams, double* axes, double* axlens, double* logvols, double log_enlarge,
                          const int* active, const double* run_shift = nullptr);
// boot.hip: the bootstrap expansion factor of runs x B resampled replicas (bounding.py:381-400, 688-703, 1593-1648);
// run_shift[run] = d * ln(expand) (0 where expand <= 1 or the run is not active); ws of bootstrap_ws_bytes(...)
size_t bootstrap_ws_bytes(int runs, int n, int d, int max_ells, int B);
int bootstrap_expand_launch(dh_ctx* ctx, int runs, const double* pts, int n, int d, int multi, int max_ells, int B,
                            const uint64_t* ent, const int* active, void* ws, double* run_shift, double* expand,
                            int* bstatus);
int eval_launch_dev(dh_ctx* ctx, int problem, int k, const double* u, double* v, double* logl);
// bound.hip: start-point membership per run (candidate w -> run w / wpr); flag[run] |= 1 if one lies outside
int contains_runs_launch(dh_ctx* ctx, const double* x, int k, int d, int wpr, const double* ctrs, const double* ams,
                         const int* nells, int max_ells, int strict, const int* run_mode, int my_mode,
                         const int* bstatus, int* flag, int* first = nullptr);
// friends.hip: Y = X M (n x d times d x d) on the context's stream, device pointers
int friends_whiten_launch(dh_ctx* ctx, const double* X, const double* M, int n, int d, double* Y);

// wide-D path (wide.hip): used by the dispatchers when the dimension exceeds the
// register-resident limits.  kind: 0 rwalk, 1 rslice, 2 slice, 3 unit cube.
int wide_walk_launch(dh_ctx* ctx, int kind, int problem, int k, int ndim, int ncdim, const double* u0,
                     const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                     int iters, int doubling, const int8_t* bc, const uint64_t* rng, double* u, double* v,
                     double* logl, int32_t* c0, int32_t* c1, int32_t* c2, int32_t* flags,
                     uint64_t* rng_out, const double* run_loglstar = nullptr, const double* run_scale = nullptr,
                     const int* run_mode = nullptr, const int* run_doubling = nullptr, int wpr = 0, int my_mode = 0,
                     const PhiloxKey* philox = nullptr);
int wide_eval_launch(dh_ctx* ctx, const ProblemDev& p, int k, const double* u, double* v, double* logl);
// UniformBoundSampler inside a (multi-)ellipsoid at wide D; problem = -1: propose only (lock-step)
int wide_unif_launch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                     const double* axes, const double* ams, const double* cumprob, double loglstar,
                     const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u, double* v, double* logl,
                     int32_t* ncalls, int32_t* flags, uint64_t* rng_out, const PhiloxKey* philox = nullptr,
                     const double* run_loglstar = nullptr, const int* run_mode = nullptr, int wpr = 1, int my_mode = 0,
                     const int* run_nells = nullptr, int run_me = 0);
int wide_contains_launch(dh_ctx* ctx, const double* x, int k, int d, const double* ctrs, const double* ams,
                         int m, int mode, int32_t* count, uint64_t* mask, double* quad);
int wide_single_launch(dh_ctx* ctx, int runs, const double* pts, int n, int d, int32_t* nells,
                       int32_t* status, double* ctrs, double* covs, double* ams, double* axes,
                       double* axlens, double* logvols);
int wide_single_launch_masked(dh_ctx* ctx, int runs, const double* pts, int n, int d, int32_t* nells,
                              int32_t* status, double* ctrs, double* covs, double* ams, double* axes,
                              double* axlens, double* logvols, const int* active);
// MultiEllipsoid.update at wide D: host recursion over device node work (wide.hip)
int wide_multi_launch(dh_ctx* ctx, int runs, const double* pts, int n, int d, int max_ells, int32_t* nells,
                      int32_t* status, double* ctrs, double* covs, double* ams, double* axes, double* axlens,
                      double* logvols, int32_t* leaf_of_point, int32_t* nnodes, const int* active = nullptr);

}  // namespace dh

#define DH_CHECK_CTX(ctx) \
  if (!(ctx)) return DH_ERR_ARG;
