// libdynhip.so: context, memory, events, problem registry (C ABI, see
// include/dynhip.h).  Kernels live in walk.hip / bound.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include "ctx.h"
#include "npy_ziggurat_tables.h"
#include "rng_pcg64.h"

static std::string g_err;

namespace dh {

int fail(dh_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_err = buf;
  return code;
}

bool hip_ok(dh_ctx* ctx, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  fail(ctx, DH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return false;
}

void arena_reset(dh_ctx* ctx) { ctx->arena_top = 0; }

int arena_reserve(dh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->arena_cap) return DH_OK;
  // growing invalidates earlier arena pointers: only legal right after reset
  if (ctx->arena_top != 0) return fail(ctx, DH_ERR_NOMEM, "arena grow while in use");
  if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync before arena grow")) return DH_ERR_HIP;
  if (ctx->arena) (void)hipFree(ctx->arena);
  ctx->arena = nullptr;
  ctx->arena_cap = 0;
  size_t cap = bytes + bytes / 2 + (1u << 20);
  if (!hip_ok(ctx, hipMalloc((void**)&ctx->arena, cap), "hipMalloc(arena)")) return DH_ERR_NOMEM;
  ctx->arena_cap = cap;
  return DH_OK;
}

void* arena_get(dh_ctx* ctx, size_t bytes) {
  size_t top = (ctx->arena_top + 255) & ~(size_t)255;
  if (top + bytes > ctx->arena_cap) {
    fail(ctx, DH_ERR_NOMEM, "arena overflow (%zu + %zu > %zu): reserve first", top, bytes,
         ctx->arena_cap);
    return nullptr;
  }
  ctx->arena_top = top + bytes;
  return ctx->arena + top;
}

bool get_problem(dh_ctx* ctx, int handle, ProblemDev* out) {
  if (handle < 0 || handle >= (int)ctx->problems.size() || !ctx->problems[handle].live) {
    fail(ctx, DH_ERR_ARG, "bad problem handle %d", handle);
    return false;
  }
  const dh_problem_rec& r = ctx->problems[handle];
  out->like_id = r.like_id;
  out->prior_id = r.prior_id;
  out->ndim = r.ndim;
  out->like_par = r.like_par;
  out->prior_par = r.prior_par;
  out->prec_t = r.prec_t;
  return true;
}

}  // namespace dh

using namespace dh;

extern "C" {

int dh_version(void) { return DH_VERSION; }

int dh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

dh_ctx* dh_create(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fail(nullptr, DH_ERR_HIP, "no HIP device visible (%s); libdynhip has no CPU fallback",
         e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= n) {
    fail(nullptr, DH_ERR_ARG, "device %d out of range (have %d)", device, n);
    return nullptr;
  }
  dh_ctx* ctx = new dh_ctx();
  ctx->device = device;
  if (const char* e = getenv("DH_COOP_LAUNCH")) ctx->coop_launch = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("DH_RWALK_FORM")) ctx->rwalk_form = atoi(e) == 1 ? 1 : atoi(e) == 2 ? 2 : 0;
  if (const char* e = getenv("DH_RWALK_ITEMS")) ctx->rwalk_items = atoi(e) != 0;
  if (const char* e = getenv("DH_RWALK_ITEMS_MB")) ctx->items_budget = (size_t)(atol(e) > 0 ? atol(e) : 1024) << 20;
  if (const char* e = getenv("DH_CUBE_FORM")) ctx->cube_form = atoi(e) == 1 ? 1 : atoi(e) == 2 ? 2 : 0;
  {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) ctx->num_cu = cu;
  }
  if (!hip_ok(ctx, hipSetDevice(device), "hipSetDevice") ||
      !hip_ok(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate") ||
      !hip_ok(ctx, hipMalloc((void**)&ctx->zig, (3 * 256 + 128) * sizeof(uint64_t)), "hipMalloc(zig)")) {
    g_err = ctx->err;
    delete ctx;
    return nullptr;
  }
  (void)hipMemcpy(ctx->zig, dh_zig_ki_host, 2048, hipMemcpyHostToDevice);
  (void)hipMemcpy(ctx->zig + 256, dh_zig_wi_bits_host, 2048, hipMemcpyHostToDevice);
  (void)hipMemcpy(ctx->zig + 512, dh_zig_fi_bits_host, 2048, hipMemcpyHostToDevice);
  {
    // PCG64 jump constants G_k = 1 + mult + ... + mult^(k-1) (mod 2^128), k = 1..64, as (hi, lo) pairs: the state k
    // steps ahead is S_0 + G_k (S_1 - S_0) (walkq.hip: one stream drawn by the 64 lanes of a wavefront)
    uint64_t jt[128];
    dh::U128 G = {0ull, 0ull};
    const dh::U128 m = {DH_PCG_MULT_HI, DH_PCG_MULT_LO}, one = {0ull, 1ull};
    for (int k = 0; k < 64; ++k) {
      G = dh::add128(dh::mul128(G, m), one);
      jt[2 * k] = G.hi;
      jt[2 * k + 1] = G.lo;
    }
    (void)hipMemcpy(ctx->zig + 768, jt, sizeof(jt), hipMemcpyHostToDevice);
  }
  if (arena_reserve(ctx, 8u << 20) != DH_OK) {
    g_err = ctx->err;
    dh_destroy(ctx);
    return nullptr;
  }
  return ctx;
}

void dh_destroy(dh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->problems) {
    if (p.like_par) (void)hipFree(p.like_par);
    if (p.prior_par) (void)hipFree(p.prior_par);
    if (p.prec_t) (void)hipFree(p.prec_t);
  }
  if (ctx->zig) (void)hipFree(ctx->zig);
  if (ctx->items) (void)hipFree(ctx->items);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->axes_t) (void)hipFree(ctx->axes_t);
  if (ctx->rebuild_ws) (void)hipFree(ctx->rebuild_ws);
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream);
    (void)hipStreamDestroy(ctx->side_stream);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_leaf) (void)hipEventDestroy(ctx->ev_leaf);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* dh_last_error(dh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int dh_sync(dh_ctx* ctx) {
  DH_CHECK_CTX(ctx);
  return hip_ok(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize") ? DH_OK : DH_ERR_HIP;
}

void* dh_stream(dh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

void* dh_malloc(dh_ctx* ctx, uint64_t bytes) {
  if (!ctx) return nullptr;
  void* p = nullptr;
  (void)hipSetDevice(ctx->device);
  if (!hip_ok(ctx, hipMalloc(&p, bytes ? bytes : 8), "hipMalloc")) return nullptr;
  return p;
}

int dh_free(dh_ctx* ctx, void* dptr) {
  DH_CHECK_CTX(ctx);
  if (!dptr) return DH_OK;
  (void)hipStreamSynchronize(ctx->stream);
  return hip_ok(ctx, hipFree(dptr), "hipFree") ? DH_OK : DH_ERR_HIP;
}

int dh_memcpy_h2d(dh_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  DH_CHECK_CTX(ctx);
  if (!hip_ok(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream), "H2D"))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_memcpy_d2h(dh_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
  DH_CHECK_CTX(ctx);
  if (!hip_ok(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H"))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_memset(dh_ctx* ctx, void* dst, int value, uint64_t bytes) {
  DH_CHECK_CTX(ctx);
  return hip_ok(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream), "memset") ? DH_OK : DH_ERR_HIP;
}

void* dh_event_create(dh_ctx* ctx) {
  if (!ctx) return nullptr;
  hipEvent_t ev;
  if (!hip_ok(ctx, hipEventCreate(&ev), "hipEventCreate")) return nullptr;
  return (void*)ev;
}

int dh_event_destroy(dh_ctx* ctx, void* ev) {
  DH_CHECK_CTX(ctx);
  return hip_ok(ctx, hipEventDestroy((hipEvent_t)ev), "hipEventDestroy") ? DH_OK : DH_ERR_HIP;
}

int dh_event_record(dh_ctx* ctx, void* ev) {
  DH_CHECK_CTX(ctx);
  return hip_ok(ctx, hipEventRecord((hipEvent_t)ev, ctx->stream), "hipEventRecord") ? DH_OK
                                                                                    : DH_ERR_HIP;
}

int dh_event_elapsed_ms(dh_ctx* ctx, void* ev0, void* ev1, double* ms) {
  DH_CHECK_CTX(ctx);
  if (!hip_ok(ctx, hipEventSynchronize((hipEvent_t)ev1), "hipEventSynchronize")) return DH_ERR_HIP;
  float f = 0.f;
  if (!hip_ok(ctx, hipEventElapsedTime(&f, (hipEvent_t)ev0, (hipEvent_t)ev1), "hipEventElapsedTime"))
    return DH_ERR_HIP;
  *ms = (double)f;
  return DH_OK;
}

int dh_problem_create(dh_ctx* ctx, int ndim, int like_id, const double* like_par, int n_like_par,
                      int prior_id, const double* prior_par, int n_prior_par) {
  DH_CHECK_CTX(ctx);
  if (ndim < 1) return fail(ctx, DH_ERR_ARG, "ndim=%d", ndim);
  int need_like = like_id == DH_LIKE_GAUSS_PREC ? 1 + ndim * ndim : 1;
  if (like_id < 0 || like_id > DH_LIKE_EGGBOX || n_like_par < need_like)
    return fail(ctx, DH_ERR_ARG, "likelihood id %d needs %d parameters, got %d", like_id, need_like,
                n_like_par);
  int need_prior = prior_id == DH_PRIOR_IDENTITY ? 0 : 2;
  if (prior_id < 0 || prior_id > DH_PRIOR_NORMAL || n_prior_par < need_prior)
    return fail(ctx, DH_ERR_ARG, "prior id %d needs %d parameters, got %d", prior_id, need_prior,
                n_prior_par);
  dh_problem_rec r;
  r.live = true;
  r.ndim = ndim;
  r.like_id = like_id;
  r.prior_id = prior_id;
  r.n_like = n_like_par;
  r.n_prior = n_prior_par;
  (void)hipSetDevice(ctx->device);
  if (!hip_ok(ctx, hipMalloc((void**)&r.like_par, sizeof(double) * (n_like_par + 1)), "hipMalloc") ||
      !hip_ok(ctx, hipMalloc((void**)&r.prior_par, sizeof(double) * (n_prior_par + 2)), "hipMalloc"))
    return DH_ERR_NOMEM;
  if (!hip_ok(ctx, hipMemcpy(r.like_par, like_par, sizeof(double) * n_like_par, hipMemcpyHostToDevice),
              "H2D like_par"))
    return DH_ERR_HIP;
  if (n_prior_par &&
      !hip_ok(ctx, hipMemcpy(r.prior_par, prior_par, sizeof(double) * n_prior_par, hipMemcpyHostToDevice),
              "H2D prior_par"))
    return DH_ERR_HIP;
  if (like_id == DH_LIKE_GAUSS_PREC && ndim <= kMaxRegDim) {
    // transposed + zero-padded copy for the column-sweep mat-vec of the walk
    // kernels (P is symmetric in exact arithmetic; transpose anyway)
    const int N = pad_dim(ndim);
    std::vector<double> pt((size_t)N * N, 0.0);
    for (int i = 0; i < ndim; ++i)
      for (int j = 0; j < ndim; ++j) pt[(size_t)j * N + i] = like_par[1 + (size_t)i * ndim + j];
    if (!hip_ok(ctx, hipMalloc((void**)&r.prec_t, sizeof(double) * N * N), "hipMalloc") ||
        !hip_ok(ctx, hipMemcpy(r.prec_t, pt.data(), sizeof(double) * N * N, hipMemcpyHostToDevice),
                "H2D prec_t"))
      return DH_ERR_HIP;
  }
  for (size_t i = 0; i < ctx->problems.size(); ++i)
    if (!ctx->problems[i].live) {
      ctx->problems[i] = r;
      return (int)i;
    }
  ctx->problems.push_back(r);
  return (int)ctx->problems.size() - 1;
}

int dh_problem_destroy(dh_ctx* ctx, int problem) {
  DH_CHECK_CTX(ctx);
  ProblemDev p;
  if (!get_problem(ctx, problem, &p)) return DH_ERR_ARG;
  (void)hipStreamSynchronize(ctx->stream);
  dh_problem_rec& r = ctx->problems[problem];
  (void)hipFree(r.like_par);
  (void)hipFree(r.prior_par);
  if (r.prec_t) (void)hipFree(r.prec_t);
  r = dh_problem_rec();
  return DH_OK;
}

}  // extern "C"
