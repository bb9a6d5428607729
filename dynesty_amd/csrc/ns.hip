// Device-resident ensemble of static nested-sampling runs (SURVEY.md section 8f-1,
// BASELINE config C5).  The run loop of the reference (sampler.py:932-1212:
// pop the worst live point, replace it by a proposal that beats it, integrate
// ln Z; :676-778 queue semantics; :625-674 bound-update policy) is executed on
// the GPU for `runs` independent runs at once; the host only enqueues a fixed
// kernel sequence per queue fill and polls a done-counter every few fills.
//
//   per fill:  ns_prepare  (policy: switch unit cube -> bound, schedule rebuilds)
//              rebuild pipeline + enlarge        (only runs that asked for it)
//              ns_select   (K start points, proposal frames, walker RNG streams)
//              unit-cube / rwalk walk kernels    (all runs x K walkers, one launch each)
//              ns_gather   (the walkers' start coordinates)
//              ns_consume  (queue consumption per run: slots sorted once, deaths walk that
//                           order; dead-point record, evidence integration, tuning)
//   at the end: ns_finish  (append the final live points, write the record)
//
// All state (live points, dead points, integrals, RNG) stays in HBM; with
// 288 GB there is room for the full dead-point history of thousands of runs.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <cmath>

#include "ctx.h"
#include "ns_sort.h"
#include "rng_pcg64.h"

using namespace dh;


namespace {

constexpr int kT = 256;
enum : int { MODE_CUBE = 0, MODE_BOUND = 1, MODE_DONE = 2, MODE_FAILED = 3, MODE_WAIT = 4 /* run_mode only: sits this fill out */ };

struct NsRun {
  double logvol, logz, h, lmax, scale, loglstar, dead_prev;
  long long it, ncall, ncall_last_update;
  int mode, need_rebuild, nbound, nfill;
  int acc, rej, doubling, due;  // due: a bound update the run is waiting for (see ns_prepare: rebuild_fill)
  double logzvar;
  uint64_t rng[4];
  long long nc_carry;  // calls of the entries popped after the last death: charged to the NEXT death (sampler.py:1141)
  // likelihood plateau (sampler.py:1112-1127, 1190-1193): deaths still to be taken with the plateau's volume
  // step (0 = not in plateau mode), ln of that step's volume, and the two sums that make var[ln Z] exact
  // although the dead points' d ln X is then no longer constant (see ns_finish)
  int pcount, sampler_failed;  // (sampler_failed: a uniform-sampler walker gave up, see ns_consume)
  // forced_exact: a start point of the run's CURRENT queue (selected, kept) lies outside its bound and the forced
  // update waits for the next fill that builds bounds; the run proposes and consumes nothing until then
  int fpend, pad_;
  double plogdvol, var_a, var_b;
};

struct NsArgs {
  int runs, nlive, ndim, K, walks, bound_multi, max_ells;
  int sampler;  // 0 rwalk, 1 rslice, 2 slice, 3 unif ; `walks` holds the slice count for 1, 2 (unused for 3)
  int bootstrap;  // replicas of the bootstrap expansion per rebuild (0: none)
  long long cap;  // dead-point capacity per run
  double dlogz, enlarge_log, facc, first_eff;
  long long first_ncall, update_interval;
  // run_nested's other stopping rules (sampler.py:1070-1093; dh_ns_set_option): maxiter / maxcall < 0 = none,
  // logl_max = +inf = none; add_live = 0: the record is the dead points' running evidence (no final live points)
  long long maxiter, maxcall;
  double logl_max;
  int add_live;
  int store_samples;
  long long* prof;   // optional (DH_NS_PROF=1): cycle counters of ns_consume's phases, run 0
  int rebuild_sync;  // 1: all bound-mode runs rebuild whenever any run is due (see ns_prepare)
  int overlap;       // 1: a run whose bound is being rebuilt sits the fill out (its rebuild runs beside the others' walk)
  int rebuild_fill;  // 1: this fill builds bounds; 0: a run that is due waits (idle) for the next fill that does
  int serial_walk;   // diagnostic (DH_NS_SERIAL=1): ns_consume walks every queue with the one-wavefront routine
  // (round 6) the slot order of every run's live points for THIS fill, sorted by the generator pass's presort workgroups
  // beside the walk (dh_ctx::PresortReq); presorted = 1: ns_consume reads it instead of sorting
  const unsigned short* presort;
  int presort_stride, presorted;
  NsRun* st;
  double* live_u;
  double* live_v;
  double* live_logl;
  double* dead_logl;
  double* dead_u;      // optional
  // queue
  double* q_u0;
  int* q_frame;
  uint64_t* q_rng;
  uint64_t* q_rng_out;
  double* r_u;
  double* r_v;
  double* r_logl;
  int* r_a;  // naccept | ncalls (unit cube, slice)
  int* r_b;  // nreject | flags (unit cube) | nexpand (slice)
  int* r_c;  // ncontract (slice)
  int* r_d;  // flags (slice)
  int* run_doubling;
  // per-run walk parameters
  double* run_loglstar;
  double* run_scale;
  int* run_mode;
  int* rebuild_mask;
  int* force;  // runs: set by the start-point membership check of the previous fill (forced rebuild, sampler.py:484-489)
  // DH_NS_OPT_FORCED_EXACT: the forced update inside the fill that found the start point
  int forced_exact;
  int* force_first;   // runs: queue index of the first start point outside the bound (INT_MAX: none)
  // (forced_exact) update_bound_if_needed runs when the queue's last entry has been POPPED, before its point replaces the
  // worst one (sampler.py:771-772, 1176-1185): when that entry made the fill's last death, the slot's previous content
  // is kept here and stands in the live set while the next regular bound is built (ns_swap_undo)
  double* undo_u;     // runs x ndim
  int* undo_slot;     // runs: the slot, or -1
  uint64_t* sel_ent;  // runs x 4: the words ns_select seeded this fill's walker selections from
  // (forced_exact) what this fill does to the run's bound: 0 nothing, 1 regular update (built before the queue is
  // selected), 2 forced update (built after: the queue found a start point outside); the runs a selection /
  // membership pass serves (MODE_BOUND / MODE_WAIT per run); which pass ns_select / ns_gather are in
  int* fx_kind;
  int* pass_mode;
  int fx_pass;  // 0: not forced_exact; 1: before the fill's rebuild (every run but the regular updates'); 2: after (those)
  int* ndone;
  // bound (rebuild outputs)
  int* nells;
  int* bstatus;
  double* b_ctrs;
  double* b_covs;
  double* b_ams;
  double* b_axes;
  double* b_axl;
  double* b_lv;
  double* b_cum;        // unif sampler: rand_choice weights of the run's ellipsoids, cumulated (bounding.py:726-731)
  uint64_t* boot_ent;   // runs x 4: the words a rebuilding run drew for its bootstrap replicas
  double* run_shift;    // runs: ndim * ln(bootstrap expansion factor) of the rebuild just done
  // results
  double* records;  // runs x 8: logz, logzerr, niter, ncall, h, nbound, status, eff
  double* fin_ws;   // runs x fin_stride: ns_finish's per-point terms (3 nlive) and, for large live sets, the sorted values
  size_t fin_stride;
  // dh_ns_consume (one queue consumption as an operator of its own): death list of THIS call
  int dead_rel;     // 1: dead_logl rows are K wide and hold this call's deaths from index 0
  int* trace_slot;  // runs x K or null: slot of every death
  int* trace_src;   // runs x K or null: queue index of its replacement
  int* trace_n;     // runs x 2 or null: number of deaths kept, stopped flag
  // per-point bookkeeping of the reference's saved_run (sampler.py:1165-1182), all optional (null together):
  int* live_it;  // runs x N   in/out: iteration at which the point living in the slot was proposed (0 = initial)
  int* dead_id;  // like dead_logl: slot of the dead point                        ('id')
  int* dead_it;  //                 iteration at which it had been proposed        ('it')
  int* dead_nc;  //                 likelihood calls spent to replace it           ('nc')
};

__device__ __forceinline__ double logaddexp_dev(double x, double y) {
  if (x == y) return x + 0.6931471805599453;
  const double d = x - y;
  if (d > 0) return x + log1p(exp(-d));
  if (d <= 0) return y + log1p(exp(d));
  return x + y;
}

// progress_integration (utils.py:1470-1492), one dead point
__device__ __forceinline__ void integrate_step(double& logz, double& h, double& logzvar, double lprev,
                                               double lnew, double logvol, double dlv) {
  const double logdvol = logvol + log(0.5 * expm1(dlv));
  const double logwt = logaddexp_dev(lnew, lprev) + logdvol;
  const double logz_new = logaddexp_dev(logz, logwt);
  const double t0 = exp(lprev - logz_new + logdvol), t1 = exp(lnew - logz_new + logdvol);
  const double lzterm = (t0 > 0.0 ? t0 * lprev : 0.0) + (t1 > 0.0 ? t1 * lnew : 0.0);
  const double w = exp(logz - logz_new);
  const double h_new = lzterm + (w > 0.0 ? w * (h + logz) : 0.0) - logz_new;
  logzvar += (h_new - h) * dlv;  // logzvar_new = logzvar + dh * dlogvol
  h = h_new;
  logz = logz_new;
}

// ---- init: RNG, uniform live points --------------------------------------
__global__ void __launch_bounds__(kT)
    ns_init(NsArgs a, const uint32_t* __restrict__ entropy, int nwords, uint32_t first_run) {
  const int run = blockIdx.x, t = threadIdx.x;
  const int N = a.nlive, D = a.ndim;
  // every live point gets its own stream: child (first_run + run) * N + i of the entropy
  for (int i = t; i < N; i += kT) {
    Pcg64 g;
    seed_from_child(g, entropy, nwords, (uint32_t)((first_run + run) * (uint32_t)N + i));
    double* u = a.live_u + ((size_t)run * N + i) * D;
    for (int j = 0; j < D; ++j) u[j] = g.next_double();
  }
  if (t == 0) {
    NsRun r;
    r.logvol = 0.0;
    r.logz = -1e300;
    r.h = 0.0;
    r.lmax = -1e300;
    r.scale = 1.0;
    r.loglstar = -1e300;
    r.dead_prev = -1e300;
    r.it = 0;
    r.ncall = N;
    r.ncall_last_update = 0;
    r.mode = MODE_CUBE;
    r.need_rebuild = 0;
    r.nbound = 0;
    r.nfill = 0;
    r.acc = r.rej = r.doubling = r.due = 0;
    r.fpend = r.pad_ = 0;
    r.logzvar = 0.0;
    r.nc_carry = 0;
    r.pcount = r.sampler_failed = 0;
    r.plogdvol = r.var_a = r.var_b = 0.0;
    Pcg64 g;
    seed_from_child(g, entropy, nwords, 0x80000000u + first_run + (uint32_t)run);
    g.store(r.rng);
    a.st[run] = r;
  }
}

// ---- wave-level minimum helpers (the queue walk of ns_consume) --------------------------------------
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  // butterflies inside each row of 16 lanes, then row_bcast15 / row_bcast31: lane 63 holds the minimum
#define DH_DPP_MIN(ctrl, rmask) \
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false))
  DH_DPP_MIN(0xB1, 0xF);   // quad_perm [1,0,3,2]
  DH_DPP_MIN(0x4E, 0xF);   // quad_perm [2,3,0,1]
  DH_DPP_MIN(0x141, 0xF);  // row_half_mirror
  DH_DPP_MIN(0x140, 0xF);  // row_mirror
  DH_DPP_MIN(0x142, 0xA);  // row_bcast:15 -> rows 1, 3
  DH_DPP_MIN(0x143, 0xC);  // row_bcast:31 -> rows 2, 3
#undef DH_DPP_MIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// lane holding the smallest of the wave's 64 keys (lowest lane on ties) and that key
__device__ __forceinline__ int wave_argmin(double key, double* kmin) {
  unsigned long long bits = (unsigned long long)__double_as_longlong(key);
  bits = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);  // order-preserving image
  const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
  const unsigned mh = wave_min_u32(hi);
  const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  const unsigned long long win = __ballot(hi == mh && lo == ml);
  const int l = __ffsll((long long)win) - 1;
  const int klo = __builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(key), l);
  const int khi = __builtin_amdgcn_readlane((int)(unsigned)(__double_as_longlong(key) >> 32), l);
  *kmin = __longlong_as_double(((long long)khi << 32) | (unsigned)klo);
  return l;
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// ---- after the initial evaluation: the worst and the best live log-likelihood of every run ------------
__global__ void __launch_bounds__(kT) ns_start(NsArgs a) {
  const int run = blockIdx.x, t = threadIdx.x, N = a.nlive;
  __shared__ double rmax[kT], rmin[kT];
  double mx = -1e300, mn = INFINITY;
  for (int i = t; i < N; i += kT) {
    const double k = a.live_logl[(size_t)run * N + i];
    mx = fmax(mx, k);
    mn = fmin(mn, k);
  }
  rmax[t] = mx;
  rmin[t] = mn;
  __syncthreads();
  for (int s = kT / 2; s > 0; s >>= 1) {
    if (t < s) {
      rmax[t] = fmax(rmax[t], rmax[t + s]);
      rmin[t] = fmin(rmin[t], rmin[t + s]);
    }
    __syncthreads();
  }
  if (t == 0) {
    a.st[run].lmax = rmax[0];
    a.st[run].loglstar = rmin[0];
  }
}

// ---- policy (sampler.py:625-674 update_bound_if_needed) ------------------------
// One workgroup looks at all runs.  With rebuild_sync the runs of the ensemble rebuild
// TOGETHER: as soon as any run is due, every run that already samples from a bound joins
// (its rebuild comes early, never late).  A rebuild is a latency-bound tree construction
// that costs about the same for 1 or 64 runs; with rwalk every run spends exactly
// K * walks calls per fill, so after the first joint rebuild all runs are due in the same
// fill and the ensemble pays one rebuild per update interval instead of one per fill.
__global__ void __launch_bounds__(kT) ns_prepare(NsArgs a) {
  const int t = threadIdx.x;
  int any = 0;
  for (int run = t; run < a.runs; run += kT) {
    NsRun& r = a.st[run];
    int want = 0;
    if (r.fpend) {
      // its queue is selected and waits for the forced update: nothing is decided for it (ns_force_prepare takes it in
      // the next fill that builds bounds)
    } else if (r.mode == MODE_CUBE || r.mode == MODE_BOUND) {
      if (r.mode == MODE_BOUND && a.bstatus[run] != DH_OK) {
        r.mode = MODE_FAILED;
        atomicAdd(a.ndone, 1);
      } else {
        const double eff = 100.0 * (double)(r.it > 0 ? r.it : 1) / (double)r.ncall;
        if (r.mode == MODE_CUBE) {
          want = (r.ncall >= a.first_ncall && eff < a.first_eff) ? 1 : 0;
        } else if (r.ncall >= r.ncall_last_update + a.update_interval || (a.force && a.force[run])) {
          // (force: a start point of the last fill lay outside the bound, sampler.py:484-489)
          want = 1;
        }
        if (r.due) want = 1;
      }
    }
    // The rebuild is a latency chain that costs about the same for one run or sixty-four, so the loop builds bounds
    // only every rebuild_every-th fill: a run that becomes due in between WAITS -- it proposes nothing and consumes
    // nothing until then.  Runs are independent, so idling changes nothing in a run's own sequence (rebuild on the
    // same live set, then walk with the same generator state): results are those of the reference schedule.
    if (want && !a.rebuild_fill) {
      r.due = 1;
      want = 0;
    }
    r.need_rebuild = want;
    any |= want;
  }
  any = __syncthreads_or(any);
  for (int run = t; run < a.runs; run += kT) {
    NsRun& r = a.st[run];
    int need = r.need_rebuild;
    if (a.rebuild_sync && any && a.rebuild_fill && r.mode == MODE_BOUND && !r.fpend) need = 1;
    if (need) {
      if (r.mode == MODE_CUBE) r.mode = MODE_BOUND;
      r.due = 0;
      r.ncall_last_update = r.ncall;
      r.nbound += 1;
      if (a.bootstrap > 0) {  // get_seed_sequence(rstate, bootstrap) (utils.py:1002-1009): one draw per rebuild
        Pcg64 g;
        g.load(r.rng);
        for (int i = 0; i < 4; ++i) a.boot_ent[(size_t)run * 4 + i] = g.next64();
        g.store(r.rng);
      }
    }
    r.need_rebuild = need;
    if (a.force) a.force[run] = 0;
    a.rebuild_mask[run] = need;
    a.run_mode[run] = (r.due || (a.overlap && need) || (r.fpend && !a.rebuild_fill)) ? MODE_WAIT : r.mode;
    if (a.fx_kind) a.fx_kind[run] = need ? 1 : 0;
    a.run_loglstar[run] = r.loglstar;
    a.run_scale[run] = r.scale;
    a.run_doubling[run] = r.doubling;
  }
  // runs still in the unit-cube phase (the host stops launching that phase's kernel once there are none:
  // a run never returns to it)
  int ncube = 0;
  for (int run = t; run < a.runs; run += kT) ncube += a.st[run].mode == MODE_CUBE ? 1 : 0;
  ncube = __syncthreads_count(ncube > 0) ? 1 : 0;
  if (t == 0) a.ndone[1] = ncube;
}

// ---- queue fill: start points, frames, walker streams ---------------------------
constexpr int kMaxCum = 4096;  // ellipsoids per run the frame choice can weigh (32 KB of LDS)
__global__ void __launch_bounds__(kT) ns_select(NsArgs a) {
  __shared__ uint64_t ent[4];
  __shared__ double cum[kMaxCum];  // rand_choice weights of ALL ellipsoids of the run (bounding.py:726-731)
  __shared__ int Msh;
  __shared__ double mxs;
  const int run = blockIdx.x, t = threadIdx.x;
  const int N = a.nlive, K = a.K;
  NsRun& r = a.st[run];
  const int mode = r.mode;
  if (mode != MODE_CUBE && mode != MODE_BOUND) return;
  if (a.run_mode[run] == MODE_WAIT) return;  // waiting for its bound: the run proposes nothing this fill
  if (a.fx_pass == 1 && (a.fx_kind[run] == 1 || r.fpend)) return;  // selects after its rebuild / keeps its queue
  if (a.fx_pass == 2 && a.fx_kind[run] != 1) return;
  if (mode == MODE_BOUND && a.bstatus[run] != DH_OK) return;  // ns_prepare fails the run next fill
  int M = 1;
  if (t == 0) {
    Pcg64 g;
    g.load(r.rng);
    for (int i = 0; i < 4; ++i) ent[i] = g.next64();
    g.store(r.rng);
    if (a.sel_ent)
      for (int i = 0; i < 4; ++i) a.sel_ent[(size_t)run * 4 + i] = ent[i];
    if (mode == MODE_BOUND && a.bound_multi) {
      M = a.nells[run];
      if (M > kMaxCum) M = kMaxCum;  // unreachable: dh_ns_ensemble refuses max_ells > kMaxCum
      double mx = -INFINITY;
      for (int e = 0; e < M; ++e) mx = fmax(mx, a.b_lv[(size_t)run * a.max_ells + e]);
      mxs = mx;
    }
    Msh = M;
  }
  __syncthreads();
  M = Msh;
  if (M > 1) {
    // cumsum(exp(logvol_ells - logsumexp)): the exponentials on all threads, the running sum by one
    for (int e = t; e < M; e += kT) cum[e] = exp(a.b_lv[(size_t)run * a.max_ells + e] - mxs);
    __syncthreads();
    if (t == 0) {
      double c = 0.0;
      for (int e = 0; e < M; ++e) {
        c += cum[e];
        cum[e] = c;
      }
      const double inv = 1.0 / c;
      for (int e = 0; e < M; ++e) cum[e] *= inv;
    }
    __syncthreads();
    if (a.sampler == 3)
      for (int e = t; e < M; e += kT) a.b_cum[(size_t)run * a.max_ells + e] = cum[e];
  }
  const double loglstar = r.loglstar;
  for (int w = t; w < K; w += kT) {
    Pcg64 g;
    U128 is = {ent[0], ent[1] + (uint64_t)w};
    U128 iq = {ent[2], ent[3] + 2ull * (uint64_t)w};
    g.seed(is, iq);
    const size_t q = (size_t)run * K + w;
    int frame = 0;
    if (mode == MODE_BOUND && a.sampler != 3) {  // (the uniform sampler starts nowhere: internal_samplers.py:214-242)
      // a live point with logl > loglstar, uniformly (sampler.py:469-474)
      int i;
      int guard = 0;
      do {
        i = (int)g.interval((uint64_t)(N - 1));
      } while (!(a.live_logl[(size_t)run * N + i] > loglstar) && ++guard < 100000);
      a.r_d[q] = i;  // the chosen live point: ns_gather copies its coordinates (r_d is free until the walkers run)
      if (M > 1) {
        // min(searchsorted(cum, xr), M - 1): first index with cum[i] >= xr
        const double xr = g.next_double();
        int lo = 0, hi = M - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (cum[mid] < xr) lo = mid + 1; else hi = mid;
        }
        frame = lo;
      }
    }
    a.q_frame[q] = run * a.max_ells + frame;
    g.store(a.q_rng + q * 4);
  }
}

// start points of the walkers: q_u0[q] = live_u[run, r_d[q]], one thread per coordinate over the whole ensemble (a
// thread per walker inside ns_select was a serial copy of D strided doubles: 0.23 ms per fill at D = 200)
__global__ void __launch_bounds__(256) ns_gather(NsArgs a) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int D = a.ndim, K = a.K, N = a.nlive;
  if (e >= (size_t)a.runs * K * D) return;
  const size_t q = e / D;
  const int j = (int)(e - q * D), run = (int)(q / K);
  if (a.st[run].mode != MODE_BOUND || a.bstatus[run] != DH_OK) return;
  if (a.run_mode[run] == MODE_WAIT) return;
  if (a.fx_pass == 1 && a.fx_kind[run] == 1) return;
  if (a.fx_pass == 2 && a.fx_kind[run] != 1) return;
  a.q_u0[e] = a.live_u[((size_t)run * N + a.r_d[q]) * D + j];
}

// ---- the forced update inside the fill (DH_NS_OPT_FORCED_EXACT; sampler.py:484-489) ----------------------------
// Sampler._fill_queue proposes the K start points one after the other; the first one that lies outside the bound
// rebuilds it at once (update_bound_if_needed(force=True): all live points, nbound + 1, the call counter of the last
// update reset), AFTER its own axes were drawn from the old bound (propose_live: get_random_axes stands before the
// membership test).  The later entries draw their axes from the new bound, which holds every live point, so there is
// at most one forced update per fill.  Here: the membership kernel has left force[run] and force_first[run];
// ns_force_prepare turns the flags into the rebuild mask and does the run's bookkeeping, ns_shadow_axes keeps the old
// frames of the masked runs in the second half of the axes array, the masked rebuild runs, and ns_reselect points the
// entries up to force_first at the kept frames and redraws the frames of the later ones from the new volumes (same
// variates: the selection generator of entry w is a function of the fill's four words and w).
// (forced_exact) the live set as the reference's regular bound update sees it: without the point that the previous
// fill's last queue entry brought in (see NsArgs::undo_u); launched before and after the masked rebuild (a swap)
__global__ void __launch_bounds__(64) ns_swap_undo(NsArgs a) {
  const int run = blockIdx.x;
  if (!a.rebuild_mask[run] || a.fx_kind[run] != 1) return;  // (a forced update sees the live set as it is)
  const int s = a.undo_slot[run];
  if (s < 0) return;
  for (int x = threadIdx.x; x < a.ndim; x += 64) {
    double* p = a.live_u + ((size_t)run * a.nlive + s) * a.ndim + x;
    double* q = a.undo_u + (size_t)run * a.ndim + x;
    const double tmp = *p;
    *p = *q;
    *q = tmp;
  }
}

// the runs a selection / membership pass serves -> pass_mode
__global__ void __launch_bounds__(kT) ns_pass_mask(NsArgs a, int pass) {
  for (int run = threadIdx.x; run < a.runs; run += kT) {
    const NsRun& r = a.st[run];
    const bool walking = r.mode == MODE_BOUND && a.run_mode[run] == MODE_BOUND && a.bstatus[run] == DH_OK;
    const int kd = a.fx_kind[run];
    a.pass_mode[run] = (walking && (pass == 1 ? kd != 1 : kd != 0)) ? MODE_BOUND : MODE_WAIT;
  }
}

// After the first membership pass.  A flagged run's forced update is built with this fill's bounds if the fill builds
// bounds; otherwise the run keeps its queue (start points, frames, streams, the selection words) and sits the fills
// out until one does -- runs are independent, so idling changes nothing in the run's own sequence (the same live set,
// the same generator state, the same queue: ns_prepare / ns_select leave a pending run alone), and the ensemble pays
// one rebuild chain per rebuild fill, as without forced updates.
__global__ void __launch_bounds__(kT) ns_force_prepare(NsArgs a) {
  for (int run = threadIdx.x; run < a.runs; run += kT) {
    NsRun& r = a.st[run];
    const int f = (a.force[run] && a.pass_mode[run] == MODE_BOUND) ? 1 : 0;
    if (f && a.rebuild_fill) {
      r.due = 0;
      r.fpend = 0;
      // update_bound_if_needed(-inf, force=True) records self.ncall (sampler.py:631-632, 674): the sampler's counter,
      // which at a refill does not yet hold the calls of the entries popped since the last death (they are still in
      // _new_point's ncall_accum, sampler.py:739-747) -- the carry
      r.ncall_last_update = r.ncall - r.nc_carry;
      r.nbound += 1;
      if (a.bootstrap > 0) {
        Pcg64 g;
        g.load(r.rng);
        for (int i = 0; i < 4; ++i) a.boot_ent[(size_t)run * 4 + i] = g.next64();
        g.store(r.rng);
      }
      a.fx_kind[run] = 2;
      a.rebuild_mask[run] = 1;
    } else if (f) {
      r.fpend = 1;
      a.run_mode[run] = MODE_WAIT;
      a.force_first[run] = 0x7fffffff;  // (found again, from the same queue and bound, in the fill that rebuilds)
    } else {
      if (a.pass_mode[run] == MODE_BOUND) {
        a.force_first[run] = 0x7fffffff;
        r.fpend = 0;  // (a pending run is flagged again by construction)
      }
    }
    a.force[run] = 0;  // (the membership test after the rebuild sets it again: update failed / a new pending run)
  }
}

__global__ void __launch_bounds__(256) ns_shadow_axes(NsArgs a) {
  const int run = blockIdx.x;
  if (!a.rebuild_mask[run] || a.fx_kind[run] != 2) return;
  const size_t dd = (size_t)a.ndim * a.ndim, n = (size_t)(a.bound_multi ? a.nells[run] : 1) * dd;
  const double* src = a.b_axes + (size_t)run * a.max_ells * dd;
  double* dst = a.b_axes + ((size_t)a.runs + run) * a.max_ells * dd;
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (size_t)gridDim.y * 256) dst[i] = src[i];
}

__global__ void __launch_bounds__(kT) ns_reselect(NsArgs a) {
  __shared__ double cum[kMaxCum];
  __shared__ double mxs;
  const int run = blockIdx.x, t = threadIdx.x;
  if (!a.rebuild_mask[run]) return;
  const int N = a.nlive, K = a.K;
  NsRun& r = a.st[run];
  if (a.fx_kind[run] == 1) {
    // a regular update whose fresh queue has a start point outside the fresh bound (the point the bound was built
    // without, sampler.py:771-772): a forced update of its own, taken in the next fill that builds bounds
    if (t == 0 && a.force[run] && a.bstatus[run] == DH_OK) {
      r.fpend = 1;
      a.run_mode[run] = MODE_WAIT;
      a.force[run] = 0;
    }
    return;
  }
  if (a.force[run] || a.bstatus[run] != DH_OK) {
    // RuntimeError('Update of the ellipsoid failed') (sampler.py:489), or the rebuild itself failed: the run proposes
    // nothing; ns_prepare ends it at the next fill
    if (t == 0) {
      if (a.bstatus[run] == DH_OK) a.bstatus[run] = DH_ERR_CONTAIN;
      a.run_mode[run] = MODE_WAIT;
      a.force_first[run] = 0x7fffffff;
    }
    return;
  }
  int M = a.bound_multi ? a.nells[run] : 1;
  if (M > kMaxCum) M = kMaxCum;
  if (M > 1) {
    if (t == 0) {
      double mx = -INFINITY;
      for (int e = 0; e < M; ++e) mx = fmax(mx, a.b_lv[(size_t)run * a.max_ells + e]);
      mxs = mx;
    }
    __syncthreads();
    for (int e = t; e < M; e += kT) cum[e] = exp(a.b_lv[(size_t)run * a.max_ells + e] - mxs);
    __syncthreads();
    if (t == 0) {
      double c = 0.0;
      for (int e = 0; e < M; ++e) {
        c += cum[e];
        cum[e] = c;
      }
      const double inv = 1.0 / c;
      for (int e = 0; e < M; ++e) cum[e] *= inv;
    }
    __syncthreads();
  }
  const int jstar = a.force_first[run];
  const double loglstar = r.loglstar;
  const uint64_t* ent = a.sel_ent + (size_t)run * 4;
  for (int w = t; w < K; w += kT) {
    const size_t q = (size_t)run * K + w;
    if (w <= jstar) {
      a.q_frame[q] += a.runs * a.max_ells;  // its axes were drawn from the old bound
      continue;
    }
    int frame = 0;
    if (M > 1) {
      Pcg64 g;
      U128 is = {ent[0], ent[1] + (uint64_t)w};
      U128 iq = {ent[2], ent[3] + 2ull * (uint64_t)w};
      g.seed(is, iq);
      int i, guard = 0;
      do {
        i = (int)g.interval((uint64_t)(N - 1));
      } while (!(a.live_logl[(size_t)run * N + i] > loglstar) && ++guard < 100000);
      const double xr = g.next_double();
      int lo = 0, hi = M - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] < xr) lo = mid + 1; else hi = mid;
      }
      frame = lo;
    }
    a.q_frame[q] = run * a.max_ells + frame;
  }
  __syncthreads();
  if (t == 0) a.force_first[run] = 0x7fffffff;
}

// ---- consume the queue (sampler.py:732-778 + 1105-1185), one workgroup per run ----
// Only the walk over the queue is inherently serial (which proposal kills which live point
// depends on every earlier one): consume_sorted, one wavefront.  Everything transcendental -- the evidence integration of progress_integration
// (utils.py:1470-1492) and the dlogz stopping rule, ~10 exp/log per dead point -- is a
// prefix scan over the deaths of the fill and runs on all lanes afterwards:
//   ln Z_e   = logaddexp-scan of the trapezoid weights,
//   lmax_e   = running maximum of the inserted log-likelihoods,
//   stop     = first e with ln(1 + exp(lmax_e + lnX_e - lnZ_e)) < dlogz,
//   H_E      = e^{-lnZ_E} [ G_0 + sum_{e<=E} dX_e (L e^L terms) ] - lnZ_E   (G = e^{lnZ}(H + lnZ) is additive),
//   var lnZ += dlnX (H_E - H_0)                         (the per-step sum telescopes: dlnX is constant).
// If the stop index falls inside the fill, the walk is replayed up to it (once per run).
#define NS_PROF(i)                                              \
  do {                                                          \
    if (a.prof && run == 0 && t == 0) {                          \
      const long long now_ = clock64();                          \
      a.prof[i] += now_ - pt_;                                   \
      pt_ = now_;                                                \
    }                                                            \
  } while (0)

constexpr int kEPTMax = 8;  // deaths per lane in the scan phases: K <= kEPTMax * kT
// (round 6) ns_consume and its parallel walk are compiled for kEPT = 1, 2, 4, 8 deaths per lane and launched for the
// smallest that holds the queue: with one form for all, a queue of 512 carried eight-element register arrays and
// eight-trip loops of which two trips did anything.

// LDS of ns_consume: keys and slot sources by slot, the sorted slot order (padded to a power of two), the queue's
// arrays and the buffer of low replacements
__host__ __device__ inline size_t ns_consume_lds_full(int N, int K) {
  size_t P = 1;
  while (P < (size_t)N) P <<= 1;
  if (P < 256) P = 256;  // (the slot order is sorted 256 entries at a time at least)
  return (size_t)N * 12 + (size_t)K * 52 + P * 2 + 64;
}
// Large live sets (round 4).  A fill makes at most K deaths, so only the K + 1 smallest live points by
// (log-likelihood, slot) can take part in it: at least one of them outlives the fill and stands before every other
// point, which therefore is never the worst.  Where the whole set does not fit the arrays above (nlive > 32 kT, the
// register sort's reach, or more than 150 KB), ns_consume first SELECTS those K + 1 from the keys in global memory
// (radix select on the order-preserving integer image of the key, ties at the threshold by lowest slot), compacts
// them in slot order -- so that "lowest slot first" among equal values is "lowest compact index first" -- and
// runs the same consumption on that subset; the slots of the death list and of the replacements are translated back
// when they are stored.  Same deaths, same replacements, same evidence as the full arrays would give
// (tests/test_gpu_ns_consume.py holds the two paths to each other and to the oracle).
// (round 6) ... and where the register sort of all slots costs more than the selection: at C3 (nlive 5 000, K = 1 024)
// the sort of 8 192 padded slots was 144 k of the kernel's 230 k cycles per fill; selected first, 1 025 of them are
// sorted.  Taken from nlive > 4 096 (the sort's 16 and 32 elements per lane) when the queue is a quarter of the live
// set or less.
__host__ __device__ inline bool ns_consume_compact(int N, int K) {
  return K + 1 < N && (N > 32 * kT || ns_consume_lds_full(N, K) > (size_t)150 * 1024 || (N > 4096 && 4 * (K + 1) <= N + 4));
}
__host__ __device__ inline bool ns_finish_big(int N) {  // ns_finish: keys by slot, keys in order, sorted slots in LDS?
  size_t P = 1;
  while (P < (size_t)N) P <<= 1;
  return N > 32 * kT || (size_t)N * 16 + (P > 256 ? P : 256) * 2 + 64 > (size_t)150 * 1024;
}
__host__ __device__ inline size_t ns_fin_stride(int N) {  // doubles of ns_finish's workspace per run
  size_t P = 1;
  while (P < (size_t)N) P <<= 1;
  return 3 * (size_t)N + (ns_finish_big(N) ? P : 0);
}
__host__ __device__ inline size_t ns_select_scratch(int N) {  // histogram, two lane masks per 64 slots, their prefixes
  const size_t nm = ((size_t)N + 63) / 64;
  return 256 * 4 + nm * 16 + nm * 8 + 64;
}
__host__ __device__ inline size_t ns_consume_lds(int N, int K) {
  if (!ns_consume_compact(N, K)) return ns_consume_lds_full(N, K);
  const size_t NC = (size_t)K + 1;
  size_t P = 1;
  while (P < NC) P <<= 1;
  if (P < 256) P = 256;
  size_t q = (size_t)K * 52, sc = ns_select_scratch(N);
  if (q < sc) q = sc;  // (the selection's scratch lies where the queue's arrays come afterwards)
  q = (q + 15) & ~(size_t)15;
  return NC * 12 + q + P * 2 + NC * 2 + 64 + 16;
}

// The serial part of the queue consumption (sampler.py:741-776), run by wave 0: stale test, death record,
// replacement.  The live points are NOT kept in a priority queue: the run's slots are sorted once per fill by
// (log-likelihood, slot) -- `sidx` -- and the worst live point is then either the next unconsumed entry of that order
// or the smallest of the replacements made during this fill.  Of those only the ones below theta = the K-th smallest
// original value can ever become the worst point within K deaths (K originals stand before any other), so they are
// the only ones kept, unsorted, in a buffer B whose minimum is re-scanned by the wave when it is consumed.  A step is
// a few LDS words instead of two sift levels of a 64-ary heap (1 950 cycles).  Ties die lowest slot first, as the
// reference's np.argmin picks them (sampler.py:1107).  Returns the number of deaths (wave-uniform); *newmin = the
// worst live log-likelihood after the walk.
struct WorstKey {
  double key;
  int slot;
};
__device__ __forceinline__ bool key_before(double ka, int sa, double kb, int sb) { return ka < kb || (ka == kb && sa < sb); }

__device__ __forceinline__ int consume_sorted(const double* skey, const unsigned short* sidx, int* src, const double* ql,
                                              double* dcur, int* dj, int* dslot, int* dsrc, double* bkey, int* bslot,
                                              int* bsrc, int N, int K, long long room, int limit, int* jcap,
                                              double* newmin, int lane) {
  int ndead = 0, ptr = 0, nb = 0, bpos = -1;
  *jcap = -1;
  const double theta = K < N ? skey[sidx[K]] : INFINITY;
  // The queue and the sorted order are walked through registers: every lane holds one of the next 64 entries, the
  // current one comes by readlane (the index is wave-uniform) -- no LDS round trip on the serial chain.
  auto lane_d = [](double v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(v), l);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(__double_as_longlong(v) >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  };
  int pb = 0;  // sorted positions pb .. pb + 63 are in the lanes
  int my_slot = pb + lane < N ? (int)sidx[pb + lane] : 0x7fffffff;
  double my_key = pb + lane < N ? skey[my_slot] : INFINITY;
  int sslot = __builtin_amdgcn_readlane(my_slot, 0);
  double skeyp = lane_d(my_key, 0), bmin = INFINITY;
  int bminslot = 0x7fffffff;
  double my_q = 0.0;
  for (int j = 0; j < K; ++j) {
    if ((j & 63) == 0) my_q = j + lane < K ? ql[j + lane] : 0.0;
    const bool from_b = nb > 0 && key_before(bmin, bminslot, skeyp, sslot);
    const double worst = from_b ? bmin : skeyp;
    const double lj = lane_d(my_q, j & 63);
    if (!(lj > worst)) continue;  // stale proposal (sampler.py:774-776)
    if (ndead >= room) {          // dead-point store exhausted
      *jcap = j;
      break;
    }
    int s, from;
    if (from_b) {
      s = bminslot;
      from = bsrc[bpos];
      // remove B[bpos] (the last entry takes its place) and find the new minimum with the whole wave
      --nb;
      wave_lds_fence();
      if (lane == 0 && bpos != nb) {
        bkey[bpos] = bkey[nb];
        bslot[bpos] = bslot[nb];
        bsrc[bpos] = bsrc[nb];
      }
      wave_lds_fence();
      double mk = INFINITY;
      int ms = 0x7fffffff, mp = -1;
      for (int q = lane; q < nb; q += 64) {
        const double kq = bkey[q];
        const int sq = bslot[q];
        if (key_before(kq, sq, mk, ms)) {
          mk = kq;
          ms = sq;
          mp = q;
        }
      }
      double kmin;
      (void)wave_argmin(mk, &kmin);
      // among the lanes holding kmin the lowest slot wins
      const unsigned cand = (nb > 0 && mk == kmin) ? (unsigned)ms : 0xFFFFFFFFu;
      const unsigned smin = wave_min_u32(cand);
      const unsigned long long win = __ballot(cand == smin && mp >= 0);
      const int wl = win ? __ffsll((long long)win) - 1 : 0;
      bpos = __builtin_amdgcn_readlane(mp, wl);
      bmin = nb > 0 ? kmin : INFINITY;
      bminslot = nb > 0 ? (int)smin : 0x7fffffff;
    } else {
      s = sslot;
      from = -1;
      ++ptr;
      if (ptr - pb == 64) {  // next 64 positions of the sorted order
        pb = ptr;
        my_slot = pb + lane < N ? (int)sidx[pb + lane] : 0x7fffffff;
        my_key = pb + lane < N ? skey[my_slot] : INFINITY;
      }
      sslot = __builtin_amdgcn_readlane(my_slot, ptr - pb);
      skeyp = lane_d(my_key, ptr - pb);
    }
    if (lane == 0) {
      dcur[ndead] = worst;
      dj[ndead] = j;
      dslot[ndead] = s;
      dsrc[ndead] = from;
      src[s] = j;
      if (lj < theta) {  // may become the worst point again within this fill
        bkey[nb] = lj;
        bslot[nb] = s;
        bsrc[nb] = j;
      }
    }
    if (lj < theta) {
      if (nb == 0 || key_before(lj, s, bmin, bminslot)) {
        bmin = lj;
        bminslot = s;
        bpos = nb;
      }
      ++nb;
    }
    ++ndead;
    if (ndead == limit) break;
  }
  wave_lds_fence();
  *newmin = (nb > 0 && key_before(bmin, bminslot, skeyp, sslot)) ? bmin : skeyp;
  return ndead;
}

// (sort_slots: ns_sort.h)
using dh_sort::key_before_nb;
using dh_sort::sort_slots;

// The same walk without the serial chain, by the whole workgroup.  Let U_j = the run's live values at the start of
// the fill plus the proposals accepted before entry j, dead or alive; c_j = the deaths before j.  Entry j is
// accepted iff q_j beats the worst live value, i.e. iff more than c_j values of U_j are below q_j:
//     R_j = #{originals < q_j} + #{accepted i < j : q_i < q_j} > c_j .
// The worst value never decreases, so a stale entry i < j (q_i <= the worst at its time) lies below every LATER
// accepted q_j: for an accepted j all j - c_j stale entries before it count into #{i < j : q_i < q_j}, and for a
// stale j at most that many do.  Hence
//     entry j is accepted  <=>  #{originals < q_j} + #{i < j : q_i < q_j}  >  j ,
// a rank that does not depend on what happened to the entries before it.  Proposals accepted after death e are
// strictly above its value, so death e is the e-th smallest of originals + all accepted proposals: a merge of the
// sorted originals with the (few) accepted values low enough to die within this fill; a proposal inherits the slot
// of the death that let it in (chains of such inheritances are followed until none is open).  Equal values are
// ordered by slot in the reference (np.argmin, sampler.py:1107), and the slot of a proposal is only known after the
// deaths below it: whenever an accepted value that could die in this fill equals an original or another such value
// (rwalk handing back its start point), and when the dead-point store runs out, this routine changes nothing and
// returns 0 -- the serial walk takes the fill.  All threads call it; cj = K ints of scratch (c_j of every entry);
// sh = 4 shared ints; on success *ndead_out / *newmin as consume_sorted.
template <int kEPT>
__device__ int consume_parallel(const double* skey, const unsigned short* sidx, int* src, const double* ql, double* dcur,
                                int* dj, int* dslot, int* dsrc, double* bkey, int* bqs, int* cj, int N, int K,
                                long long room, int* sh, int* ndead_out, double* newmin, long long* prof) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  long long pt_ = prof ? clock64() : 0;
#define CP_PROF(i)                         \
  do {                                     \
    if (prof && t == 0) {                  \
      const long long now_ = clock64();    \
      prof[i] += now_ - pt_;               \
      pt_ = now_;                          \
    }                                      \
  } while (0)
  __shared__ int wcnt[kEPT][kT / 64];
  int* rA = dsrc;                  // #{originals < q_j}   (dsrc is written last)
  int* posB = (int*)dcur;          // death index of the r-th lowest accepted value   (dcur likewise)
  int* bq = posB + K;              // its queue entry
  double qv[kEPT];
#pragma unroll
  for (int u = 0; u < kEPT; ++u) {
    const int j = t + u * kT;
    qv[u] = j < K ? ql[j] : -INFINITY;
    if (j < K) {
      int lo = 0, hi = N;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (skey[sidx[mid]] < qv[u]) lo = mid + 1; else hi = mid;
      }
      rA[j] = lo;
    }
  }
  CP_PROF(11);
  if (t == 0) sh[0] = sh[1] = sh[2] = sh[3] = 0;  // [0] low accepted values, [1] tie, [2] undecided entries, [3] give up
  __syncthreads();
  // #{i < j : q_i < q_j} is needed only where the originals alone do not decide (rA[j] <= j: a quarter of a late
  // fill): those entries are listed, and every one is counted by four lanes (a quarter of i < j each)
  const int nu = (K + kT - 1) / kT;
  int* needl = posB;  // (scratch until the merge)
  int* lcnt = bq;
  for (int u = 0; u < nu; ++u) {
    const int j = t + u * kT;
    const bool need = j < K && rA[j] <= j;
    const unsigned long long b = __ballot(need);
    if (j < K) cj[j] = 1;  // accepted unless found otherwise below
    if (b) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&sh[2], __popcll(b));
      base = __builtin_amdgcn_readfirstlane(base);
      if (need) {
        const int x = base + __popcll(b & ((1ull << lane) - 1ull));
        needl[x] = j;
        lcnt[x] = 0;
      }
    }
  }
  __syncthreads();
  const int nn = sh[2];
  for (int w = t; w < 4 * nn; w += kT) {
    const int x = w >> 2, part = w & 3, j = needl[x];
    const double q = ql[j];
    int c = 0, i = (j * part) >> 2;
    const int i1 = (j * (part + 1)) >> 2;
    for (; i + 8 <= i1; i += 8) {  // (eight loads in flight: the loop is LDS latency otherwise)
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ql[i + k];
#pragma unroll
      for (int k = 0; k < 8; ++k) c += v[k] < q ? 1 : 0;
    }
    for (; i < i1; ++i) c += ql[i] < q ? 1 : 0;
    if (c) atomicAdd(&lcnt[x], c);
  }
  __syncthreads();
  for (int x = t; x < nn; x += kT) {
    const int j = needl[x];
    cj[j] = rA[j] + lcnt[x] > j ? 1 : 0;
  }
  __syncthreads();
  CP_PROF(12);
  bool acc[kEPT];
  int pre[kEPT];
#pragma unroll
  for (int u = 0; u < kEPT; ++u) {
    const int j = t + u * kT;
    acc[u] = u < nu && j < K && cj[j] != 0;
    pre[u] = 0;
    if (u < nu) {
      const unsigned long long b = __ballot(acc[u]);
      pre[u] = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[u][wv] = __popcll(b);
    }
  }
  __syncthreads();
  int ndead = 0;
  {
    int run_sum = 0;
    for (int uu = 0; uu < nu; ++uu)
      for (int w = 0; w < kT / 64; ++w) {
        const int c = wcnt[uu][w];
#pragma unroll
        for (int u = 0; u < kEPT; ++u)
          if (uu == u && w == wv) pre[u] += run_sum;
        run_sum += c;
      }
    ndead = run_sum;
  }
  if ((long long)ndead > room) return 0;  // (uniform)
#pragma unroll
  for (int u = 0; u < kEPT; ++u) {
    const int j = t + u * kT;
    if (u < nu) {
      const bool low = acc[u] && rA[j] <= ndead;  // may die within this fill (or be the worst point left)
      const unsigned long long b = __ballot(low);
      int base = 0;
      if (b) {
        if (lane == 0) base = atomicAdd(&sh[0], __popcll(b));
        base = __builtin_amdgcn_readfirstlane(base);
      }
      if (j < K) {
        cj[j] = pre[u];
        if (acc[u]) dj[pre[u]] = j;
        if (low) {
          const int x = base + __popcll(b & ((1ull << lane) - 1ull));
          bkey[x] = qv[u];
          bq[x] = j;
          if (rA[j] < N && skey[sidx[rA[j]]] == qv[u]) sh[1] = 1;  // ties with an original: slot order decides (below)
        }
      }
    }
  }
  __syncthreads();
  const int nB = sh[0];
  CP_PROF(13);
  // rank among the low accepted values (counted: there are few)
  int myr[kEPT], myq[kEPT];
  for (int u = 0; u < kEPT; ++u) {
    const int x = t + u * kT;
    myr[u] = -1;
    if (x < nB) {
      const double v = bkey[x];
      int r = 0, eq = 0;
      for (int y = 0; y < nB; ++y) {
        const double w = bkey[y];
        r += (w < v || (w == v && y < x)) ? 1 : 0;  // equal values: a provisional order, settled by slot below
        eq += w == v ? 1 : 0;
      }
      if (eq > 1) sh[1] = 1;
      if (eq > 8) sh[3] = 1;  // (a plateau of proposals: left to the serial walk)
      myr[u] = r;
      myq[u] = bq[x];
    }
  }
  __syncthreads();
  if (sh[3]) return 0;
  const bool tied = sh[1] != 0;  // an accepted value that could die now equals an original or another such value
  for (int u = 0; u < kEPT; ++u)
    if (myr[u] >= 0) {
      posB[myr[u]] = rA[myq[u]] + myr[u];
      bq[myr[u]] = myq[u];  // (ranks are a permutation; every old bq entry sits in a register by now)
      bqs[myr[u]] = myq[u];  // ... and once more where the death list will not overwrite it
    }
  __syncthreads();
  CP_PROF(14);
  // death e = the element of rank e in the merge; e == ndead: the worst point left
  double dval[kEPT + 1];
  int dsl[kEPT + 1], dsr[kEPT + 1];
  for (int u = 0; u <= kEPT; ++u) {
    const int e = u < kEPT ? t + u * kT : (t == 0 ? ndead : ndead + 1);  // (thread 0: also the worst point left)
    dsl[u] = dsr[u] = -1;
    dval[u] = INFINITY;
    if (e <= ndead) {
      int lo = 0, hi = nB;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (posB[mid] < e) lo = mid + 1; else hi = mid;
      }
      if (lo < nB && posB[lo] == e) {
        dsr[u] = bq[lo];
        dval[u] = ql[dsr[u]];
      } else if (e - lo < N) {
        dsl[u] = sidx[e - lo];
        dval[u] = skey[dsl[u]];
      }
    }
  }
  __syncthreads();
  for (int u = 0; u < kEPT; ++u) {
    const int e = t + u * kT;
    if (e < ndead) {
      dcur[e] = dval[u];
      dsrc[e] = dsr[u];
      dslot[e] = dsl[u];
    }
  }
  if (t == 0) {
    *newmin = dval[kEPT];
    *ndead_out = ndead;
  }
  __syncthreads();
  CP_PROF(15);
  // Equal values die lowest slot first (np.argmin, sampler.py:1107), and a proposal's slot is that of the death that
  // let it in -- a death of a strictly lower value.  The merge above put equal values in a provisional order; the
  // positions of a group of equal values that holds a proposal are marked (-2) and settled -- group by group, from the
  // lowest value up, as their members' slots become known -- by selecting the (e - group start)-th smallest slot of
  // the group: its originals (already in slot order) and its proposals (at most 8).
  auto val_of = [&](int r) { return ql[bqs[r]]; };
  if (tied) {
    for (int e = t; e < ndead; e += kT) {
      const double v = dcur[e];
      int lo = 0, hi = nB;  // proposals of value v: [rl, rh)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (val_of(mid) < v) lo = mid + 1; else hi = mid;
      }
      const int rl = lo;
      hi = nB;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (val_of(mid) <= v) lo = mid + 1; else hi = mid;
      }
      const int rh = lo;
      if (rh > rl) {
        const bool orig_too = dsrc[e] < 0 || (e + 1 < ndead && dcur[e + 1] == v && dsrc[e + 1] < 0) ||
                              (e > 0 && dcur[e - 1] == v && dsrc[e - 1] < 0);
        // (an original of the same value sits, in the provisional order, right behind the group's proposals; one
        // that is not among the deaths still counts: test the sorted originals)
        int pl = 0, ph = N;
        while (pl < ph) {
          const int mid = (pl + ph) >> 1;
          if (skey[sidx[mid]] < v) pl = mid + 1; else ph = mid;
        }
        const bool has_orig = orig_too || (pl < N && skey[sidx[pl]] == v);
        if (rh - rl > 1 || has_orig) dslot[e] = -2;
      }
    }
    __syncthreads();
  }
  for (;;) {
    int open = 0;
    for (int e = t; e < ndead; e += kT) {
      const int s0 = ((volatile int*)dslot)[e];
      if (s0 == -1) {  // a proposal alone at its value: the slot of the death that let it in
        const int sl = ((volatile int*)dslot)[cj[dsrc[e]]];
        if (sl >= 0)
          ((volatile int*)dslot)[e] = sl;
        else
          open = 1;
      } else if (s0 == -2) {
        const double v = dcur[e];
        int lo = 0, hi = nB;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (val_of(mid) < v) lo = mid + 1; else hi = mid;
        }
        const int rl = lo;
        hi = nB;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (val_of(mid) <= v) lo = mid + 1; else hi = mid;
        }
        const int rh = lo, gi = rh - rl;
        int pl = 0, ph = N;
        while (pl < ph) {
          const int mid = (pl + ph) >> 1;
          if (skey[sidx[mid]] < v) pl = mid + 1; else ph = mid;
        }
        int pu = pl, pe = N;
        while (pu < pe) {
          const int mid = (pu + pe) >> 1;
          if (skey[sidx[mid]] <= v) pu = mid + 1; else pe = mid;
        }
        // the proposals' slots (all must be known)
        int sr[8];
        bool known = true;
        for (int q = 0; q < 8; ++q) {
          sr[q] = 0x7fffffff;
          if (q < gi) {
            sr[q] = ((volatile int*)dslot)[cj[bqs[rl + q]]];
            known = known && sr[q] >= 0;
          }
        }
        if (!known) {
          open = 1;
        } else {
          const int m = e - (pl + rl);  // rank inside the group
          int chosen = -1, below = 0;
          for (int q = 0; q < 8; ++q)
            if (q < gi) {
              int rk = 0;
              {  // originals of the group with a smaller slot (they are in slot order)
                int a0 = pl, a1 = pu;
                while (a0 < a1) {
                  const int mid = (a0 + a1) >> 1;
                  if ((int)sidx[mid] < sr[q]) a0 = mid + 1; else a1 = mid;
                }
                rk = a0 - pl;
              }
              for (int q2 = 0; q2 < 8; ++q2) rk += (q2 < gi && sr[q2] < sr[q]) ? 1 : 0;
              if (rk == m) chosen = q;
              below += rk < m ? 1 : 0;
            }
          if (chosen >= 0) {
            dsrc[e] = bqs[rl + chosen];
            ((volatile int*)dslot)[e] = sr[chosen];
          } else {
            dsrc[e] = -1;
            ((volatile int*)dslot)[e] = (int)sidx[pl + m - below];
          }
        }
      }
    }
    if (!__syncthreads_or(open)) break;
  }
  for (int j = t; j < K; j += kT)
    if (cj[j] < ndead && dj[cj[j]] == j) atomicMax(&src[dslot[cj[j]]], j);
  __syncthreads();
  CP_PROF(7);
  return 1;
}

template <int kEPT>
__global__ void __launch_bounds__(kT) ns_consume(NsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int run = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int N = a.nlive, D = a.ndim, K = a.K;
  NsRun& r = a.st[run];
  const int mode = r.mode;
  if (mode != MODE_CUBE && mode != MODE_BOUND) return;
  if (a.run_mode && a.run_mode[run] == MODE_WAIT) return;
  if (mode == MODE_BOUND && a.bstatus[run] != DH_OK) return;
  long long pt_ = a.prof ? clock64() : 0;
  // NC = the slots the consumption works on: all of them, or the K + 1 smallest in slot order (ns_consume_compact)
  const bool compact = ns_consume_compact(N, K);
  const int NC = compact ? K + 1 : N;
  int P = 1;
  while (P < NC) P <<= 1;
  size_t qbytes = (size_t)K * 52;
  if (compact) {
    const size_t sc = ns_select_scratch(N);
    if (qbytes < sc) qbytes = sc;
    qbytes = (qbytes + 15) & ~(size_t)15;
  }
  double* skey = (double*)smem;     // NC  live log-likelihoods by (compact) slot
  double* ql = skey + NC;           // K   proposal logl
  double* dcur = ql + K;            // K   death list: logl of the dead point
  double* bkey = dcur + K;          // K   low replacements of this fill (see consume_sorted)
  int* qc = (int*)(bkey + K);       // K   calls
  int* dj = qc + K;                 // K   death list: queue index of the replacement
  int* dslot = dj + K;              // K               slot
  int* dsrc = dslot + K;            // K               content source at death
  int* qborn = dsrc + K;            // K   (per-point bookkeeping only) death index at which entry j went live
  int* bslot = qborn + K;           // K
  int* bsrc = bslot + K;            // K
  int* src = (int*)((unsigned char*)ql + qbytes);      // NC  queue index now living in the slot, -1 = original
  unsigned short* sidx = (unsigned short*)(src + NC);  // P   slots in ascending (logl, slot) order; 0xFFFF = padding
  unsigned short* cslot = sidx + (P > kT ? P : kT);    // NC  (compact) the live slot behind a compact index
  __shared__ int misc[8];
  __shared__ double wred[2][4];
  __shared__ double bcast[4];
  __shared__ long long lred[4];
  __shared__ int sel_i[4];
  __shared__ unsigned long long sel_u[2];
  double sel_v = INFINITY;  // compact: the threshold value (the (K + 1)-th smallest key) ...
  int sel_extra = 0;        // ... and how many live points OUTSIDE the subset carry it
  if (!compact) {
    // (round 6: eight keys of a thread in flight -- one load, one LDS store at a time the 2 000 keys of a C2 run were
    // eight cold round trips in a row, 11 k cycles of every fill)
    const double* keys = a.live_logl + (size_t)run * N;
    for (int i0 = t; i0 < N; i0 += 8 * kT) {
      double kv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) kv[q] = keys[i0 + q * kT < N ? i0 + q * kT : 0];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q * kT < N) {
          skey[i0 + q * kT] = kv[q];
          src[i0 + q * kT] = -1;
        }
    }
  } else {
    const double* keys = a.live_logl + (size_t)run * N;
    int* hist = (int*)ql;                                            // 256
    const int nm = (N + 63) >> 6;
    unsigned long long* lmask = (unsigned long long*)(hist + 256);   // nm: lanes below the threshold
    unsigned long long* tmask = lmask + nm;                          // nm: lanes at the threshold
    int* pless = (int*)(tmask + nm);                                 // nm: exclusive prefixes of the two counts
    int* ptie = pless + nm;
    // the order-preserving image of a key (-0 folded into +0, as the double comparison has it)
    auto ord = [](double x) -> unsigned long long {
      const unsigned long long b = (unsigned long long)__double_as_longlong(x + 0.0);
      return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    };
    unsigned long long prefix = 0, mask = 0;
    int want = K + 1;  // rank (1-based) of the threshold among the keys that share the prefix
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      hist[t] = 0;
      __syncthreads();
      // (eight independent loads in flight per thread: one at a time a pass is N / kT global round trips)
      for (int i0 = t; i0 < N; i0 += 8 * kT) {
        double kv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) kv[q] = i0 + q * kT < N ? keys[i0 + q * kT] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const unsigned long long o = ord(kv[q]);
          if (i0 + q * kT < N && (o & mask) == prefix) atomicAdd(&hist[(int)((o >> shift) & 255ull)], 1);
        }
      }
      __syncthreads();
      if (wv == 0) {
        // first digit whose cumulative count reaches `want`: a wave scan over four bins per lane (one thread walking
        // the 256 bins was 256 dependent LDS reads per pass)
        const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const int sum4 = h0 + h1 + h2 + h3;
        int incl = sum4;
        for (int off = 1; off < 64; off <<= 1) {
          const int y = __shfl_up(incl, off);
          if (lane >= off) incl += y;
        }
        const unsigned long long reach = __ballot(incl >= want);
        const int hitl = reach ? (int)__ffsll((long long)reach) - 1 : 63;
        if (lane == hitl) {
          int cum = incl - sum4, d = 4 * lane;
          const int hh[4] = {h0, h1, h2, h3};
          int q = 0;
          while (q < 3 && cum + hh[q] < want) cum += hh[q++];
          d += q;
          sel_i[0] = d;
          sel_i[1] = want - cum;
          sel_i[2] = hh[q];
        }
      }
      __syncthreads();
      prefix |= (unsigned long long)sel_i[0] << shift;
      mask |= 255ull << shift;
      want = sel_i[1];
      __syncthreads();
    }
    // `want` of the keys equal to the threshold belong to the subset: the lowest slots
    const int need = want, eq_total = sel_i[2];
    sel_extra = eq_total - need;
    for (int m0 = wv; m0 < nm; m0 += 4 * (kT / 64)) {
      double kv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (m0 + q * (kT / 64)) * 64 + lane;
        kv[q] = i < N ? keys[i] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mi = m0 + q * (kT / 64), i = mi * 64 + lane;
        const unsigned long long o = ord(kv[q]);
        const unsigned long long lm = __ballot(i < N && o < prefix), tm = __ballot(i < N && o == prefix);
        if (lane == 0 && mi < nm) {
          lmask[mi] = lm;
          tmask[mi] = tm;
        }
      }
    }
    __syncthreads();
    if (wv == 0) {  // exclusive prefixes of the per-mask counts, one wavefront
      int bl = 0, bt = 0;
      for (int m0 = 0; m0 < nm; m0 += 64) {
        const int mi = m0 + lane;
        const int cl = mi < nm ? __popcll(lmask[mi]) : 0, ct = mi < nm ? __popcll(tmask[mi]) : 0;
        int sl = cl, st = ct;
        for (int off = 1; off < 64; off <<= 1) {
          const int yl = __shfl_up(sl, off), yt = __shfl_up(st, off);
          if (lane >= off) {
            sl += yl;
            st += yt;
          }
        }
        if (mi < nm) {
          pless[mi] = bl + sl - cl;
          ptie[mi] = bt + st - ct;
        }
        bl += __shfl(sl, 63);
        bt += __shfl(st, 63);
      }
    }
    __syncthreads();
    for (int mi = wv; mi < nm; mi += kT / 64) {
      const int i = mi * 64 + lane;
      const unsigned long long lm = lmask[mi], tm = tmask[mi], below = (1ull << lane) - 1ull;
      const int tb = ptie[mi] + __popcll(tm & below);
      const bool take = ((lm >> lane) & 1ull) || (((tm >> lane) & 1ull) && tb < need);
      if (take) {
        const int pos = pless[mi] + __popcll(lm & below) + (tb < need ? tb : need);
        skey[pos] = keys[i];
        cslot[pos] = (unsigned short)i;
        src[pos] = -1;
      }
      if (((tm >> lane) & 1ull) && tb == 0) sel_u[0] = (unsigned long long)__double_as_longlong(keys[i]);
    }
    __syncthreads();
    sel_v = __longlong_as_double((long long)sel_u[0]);
    __syncthreads();  // (the scratch is the queue's from here on)
  }
  auto real_slot = [&](int sl) -> int { return compact ? (int)cslot[sl] : sl; };
  __syncthreads();
  NS_PROF(16);
  if (a.presorted && !compact) {
    const int nsidx = P > kT ? P : kT;
    const unsigned short* pre = a.presort + (size_t)run * a.presort_stride;
    for (int i0 = t; i0 < nsidx; i0 += 8 * kT) {
      unsigned short sv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) sv[q] = pre[i0 + q * kT < nsidx ? i0 + q * kT : 0];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q * kT < nsidx) sidx[i0 + q * kT] = sv[q];
    }
    __syncthreads();
  } else {
    const int nsidx = P > kT ? P : kT;
    switch (nsidx / kT) {
      case 1: sort_slots<1>(skey, sidx, NC, nsidx); break;
      case 2: sort_slots<2>(skey, sidx, NC, nsidx); break;
      case 4: sort_slots<4>(skey, sidx, NC, nsidx); break;
      case 8: sort_slots<8>(skey, sidx, NC, nsidx); break;
      case 16: sort_slots<16>(skey, sidx, NC, nsidx); break;
      default: sort_slots<32>(skey, sidx, NC, nsidx); break;
    }
  }
  NS_PROF(17);
  int acc = 0, rej = 0;
  for (int j = t; j < K; j += kT) {
    const size_t q = (size_t)run * K + j;
    ql[j] = a.r_logl[q];
    if (mode == MODE_CUBE) {
      qc[j] = a.r_a[q];
    } else if (a.sampler == 3) {
      qc[j] = a.r_a[q];
      if (a.r_b[q] & 3) ql[j] = -INFINITY;  // no ellipsoid held the draw / the try limit: never accepted, the run fails below
      if (a.r_b[q] & 3) atomicOr(&r.sampler_failed, 1);
    } else if (a.sampler == 0) {
      qc[j] = a.walks;
      acc += a.r_a[q];
      rej += a.r_b[q];
      // No step accepted: the walker hands back its start point, a LIVE point, and the reference's re-evaluation of it
      // (internal_samplers.py:970-975) gives that point's own ln L again, bit for bit -- an exact tie, which dies
      // lowest slot first and opens the plateau mode (sampler.py:1107-1119).  Two kernels evaluating the same point
      // can differ in the last bit (summation order), so the live point's stored value is taken.
      if (a.r_a[q] == 0) ql[j] = a.live_logl[(size_t)run * N + a.r_d[q]];
    } else {
      qc[j] = a.r_a[q];
      acc += a.r_b[q];                      // n_expand
      rej += a.r_c[q];                      // n_contract
      if (a.r_d[q] & 1) atomicOr(&r.doubling, 1);  // expansion_warning_set -> slice_doubling
      if (a.r_d[q] & 2) ql[j] = -INFINITY;         // failed slice: never accepted
    }
  }
  // block sums of accept / reject for the scale tuning
  __shared__ int racc[kT], rrej[kT];
  racc[t] = acc;
  rrej[t] = rej;
  if (t == 0) misc[3] = 0x7fffffff;  // first stop index
  __syncthreads();
  for (int sft = kT / 2; sft > 0; sft >>= 1) {
    if (t < sft) {
      racc[t] += racc[t + sft];
      rrej[t] += rrej[t + sft];
    }
    __syncthreads();
  }
  const long long it0 = r.it, carry0 = r.nc_carry, ncall0 = r.ncall;
  const double logvol0 = r.logvol, logz0 = r.logz, h0 = r.h, lmax0 = r.lmax, dead_prev0 = r.dead_prev;
  const double dlv = log(((double)N + 1.0) / (double)N);
  const double ldv_c = log(0.5 * expm1(dlv));  // ln(dX_e / X_e) of the trapezoid rule
  NS_PROF(0);
  // ---- phase A: the walk over the queue (wave 0) ----
  const long long room = a.dead_rel ? (long long)K + 1 : a.cap - it0;
  int walked = 0;
  if (!a.serial_walk) {
    walked = consume_parallel<kEPT>(skey, sidx, src, ql, dcur, dj, dslot, dsrc, bkey, bslot, qborn, NC, K, room, &misc[4],
                              &misc[0], &bcast[2], run == 0 ? a.prof : nullptr);
    if (walked && t == 0) misc[1] = -1;
    if (a.prof && run == 0 && t == 0 && !walked) atomicAdd((unsigned long long*)&a.prof[10], 1ull);
  }
  if (!walked && t < 64) {
    int jcap;
    double nm;
    const int nd = consume_sorted(skey, sidx, src, ql, dcur, dj, dslot, dsrc, bkey, bslot, bsrc, NC, K, room, K + 1,
                                  &jcap, &nm, t);
    if (t == 0) {
      misc[0] = nd;
      misc[1] = jcap;
      bcast[2] = nm;
    }
  }
  __syncthreads();
  const int ndead = misc[0], jcap = misc[1];
  NS_PROF(1);
  // ---- likelihood plateaus (sampler.py:1112-1127, 1190-1193) ----
  // When the worst live point shares its log-likelihood with others -- rwalk hands back its start point when no
  // step was accepted: about one queue fill in seven of a C2 run meets such a pair -- the reference takes the
  // next `multiplicity` deaths with a constant VOLUME step X_s / (N + 1) instead of the constant ln X step.
  // Equal values die consecutively, so the plateaus of a fill are its runs of equal death values (and what an
  // earlier fill left open); the j-th death of a plateau shrinks ln X by ln((N + 2 - j) / (N + 1 - j)) whatever
  // X_s was.  Every death's own step is therefore known without walking the fill; a sum scan of the steps gives
  // ln X, and the scans below take both per death (has_tie) instead of the closed form of the constant step.
  __shared__ int pl_int[4];
  __shared__ double pl_dbl[2];
  __shared__ double sred[3][4];
  int has_tie = r.pcount > 0 ? 1 : 0;
  for (int e = t; e < ndead; e += kT)
    if ((e + 1 < ndead && dcur[e + 1] == dcur[e]) || (e + 1 == ndead && bcast[2] == dcur[e])) has_tie = 1;
  has_tie = __syncthreads_or(has_tie);
  if (a.prof && t == 0 && has_tie) atomicAdd((unsigned long long*)&a.prof[8 + (r.pcount > 0 ? 1 : 0)], 1ull);
  int ntail_tot = 0;
  if (has_tie) {
    // live points left after this fill's deaths that share the last death's value (a plateau that goes on)
    int ntail = 0;
    if (ndead > 0 && bcast[2] == dcur[ndead - 1]) {
      const double x = dcur[ndead - 1];
      for (int sl = t; sl < NC; sl += kT) ntail += (src[sl] >= 0 ? ql[src[sl]] : skey[sl]) == x ? 1 : 0;
      // (compact: no death value exceeds the threshold, and the points left out are not below it)
      if (t == 0 && x == sel_v) ntail += sel_extra;
    }
    if (t == 0) pl_int[2] = 0;
    __syncthreads();
    if (ntail) atomicAdd(&pl_int[2], ntail);
    __syncthreads();
    ntail_tot = pl_int[2];
  }
  const int pc_in = r.pcount;                      // deaths an open plateau still has to take
  const int pc0 = pc_in < ndead ? pc_in : ndead;   // ... of which this fill holds the first pc0
  const double rho0 = pc_in > 0 ? exp(r.plogdvol - logvol0) : 0.0;
  // position j (1-based) of death e >= pc0 in its run of equal values and the run's multiplicity m at its start
  auto run_info = [&](int e, int& j, int& m) {
    const double x = dcur[e];
    int sst = e, fin = e;
    while (sst > pc0 && dcur[sst - 1] == x) --sst;
    while (fin + 1 < ndead && dcur[fin + 1] == x) ++fin;
    m = fin - sst + 1 + ((fin == ndead - 1 && bcast[2] == x) ? ntail_tot : 0);
    j = e - sst + 1;
  };
  auto step_of = [&](int e) -> double {  // -d ln X of death e
    if (e < pc0) return -log1p(-rho0 / (1.0 - (double)e * rho0));
    int j, m;
    run_info(e, j, m);
    return m > 1 ? -log1p(-1.0 / ((double)N + 2.0 - (double)j)) : dlv;
  };
  // sum over the workgroup's per-thread totals: returns this thread's exclusive prefix (threads own contiguous deaths)
  auto excl_sum = [&](double v, double* scratch4) -> double {
    double sc = v;
    for (int off = 1; off < 64; off <<= 1) {
      const double y = __shfl_up(sc, off);
      if (lane >= off) sc += y;
    }
    double ex = __shfl_up(sc, 1);
    if (lane == 0) ex = 0.0;
    __syncthreads();
    if (lane == 63) scratch4[wv] = sc;
    __syncthreads();
    double pre = 0.0;
    for (int w2 = 0; w2 < wv; ++w2) pre += scratch4[w2];
    return pre + ex;
  };
  NS_PROF(18);
  // ---- phase B: integration + stopping rule as prefix scans over the deaths ----
  const int EPT = (K + kT - 1) / kT;
  double lw[kEPT], nl[kEPT], cd[kEPT], lvv[kEPT];
  {
    double run_cd = 0.0;
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      const int e = t * EPT + i;
      cd[i] = dlv;
      lvv[i] = logvol0 - (double)(e + 1) * dlv;
      if (has_tie && i < EPT && e < ndead) {
        cd[i] = step_of(e);
        run_cd += cd[i];
        lvv[i] = run_cd;
      }
    }
    if (has_tie) {
      const double ex = excl_sum(run_cd, sred[0]);
#pragma unroll
      for (int i = 0; i < kEPT; ++i) lvv[i] = logvol0 - (ex + lvv[i]);
    }
  }
  // ln Z after every death = ln(Z_0 + prefix sum of the weights): the weights are summed in the LINEAR domain, relative
  // to M = the largest of them and of ln Z_0 (one exp and one log per death; a logaddexp scan -- an exp and a log1p
  // for every element AND every scan step -- was the largest part of this phase: 41 k of its 67 k cycles at C2).  The
  // recurrence's rounding differs from the serial chain's in the last bits only (tested to 1e-10 absolute).
  double mloc = logz0;
#pragma unroll
  for (int i = 0; i < kEPT; ++i) {
    lw[i] = -INFINITY;
    nl[i] = -INFINITY;
    const int e = t * EPT + i;
    if (i < EPT && e < ndead) {
      const double lnew = dcur[e], lprev = e ? dcur[e - 1] : dead_prev0;
      const double logvol_e = lvv[i];
      lw[i] = logaddexp_dev(lnew, lprev) + logvol_e + (has_tie ? log(0.5 * expm1(cd[i])) : ldv_c);
      nl[i] = ql[dj[e]];
      mloc = fmax(mloc, lw[i]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) mloc = fmax(mloc, __shfl_xor(mloc, off));
  if (lane == 0) wred[0][wv] = mloc;
  __syncthreads();
  const double M = fmax(fmax(wred[0][0], wred[0][1]), fmax(wred[0][2], wred[0][3]));
  __syncthreads();
  double tz = 0.0, tm = -INFINITY;
#pragma unroll
  for (int i = 0; i < kEPT; ++i)
    if (i < EPT) {
      tz += lw[i] > -INFINITY ? exp(lw[i] - M) : 0.0;  // running (inclusive) values of this lane's segment
      tm = fmax(tm, nl[i]);
      lw[i] = tz;
      nl[i] = tm;
    }
  // inclusive scan of the lane totals across the wave, then across the 4 waves
  double sz = tz, sm = tm;
  for (int off = 1; off < 64; off <<= 1) {
    const double yz = __shfl_up(sz, off), ym = __shfl_up(sm, off);
    if (lane >= off) {
      sz += yz;
      sm = fmax(ym, sm);
    }
  }
  if (lane == 63) {
    wred[0][wv] = sz;
    wred[1][wv] = sm;
  }
  double ez = __shfl_up(sz, 1), em = __shfl_up(sm, 1);  // exclusive prefix inside the wave
  if (lane == 0) {
    ez = 0.0;
    em = -INFINITY;
  }
  __syncthreads();
  const double z0 = logz0 > -INFINITY ? exp(logz0 - M) : 0.0;
  double sbefore = z0, pm = lmax0;  // everything before this lane's segment
  for (int w2 = 0; w2 < wv; ++w2) {
    sbefore += wred[0][w2];
    pm = fmax(pm, wred[1][w2]);
  }
  sbefore += ez;
  pm = fmax(pm, em);
  const double pz = M + log(sbefore);  // ln Z before this lane's segment
  NS_PROF(19);
  int mystop = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < kEPT; ++i) {
    const int e = t * EPT + i;
    if (i < EPT && e < ndead) {
      lw[i] = M + log(sbefore + lw[i]);  // ln Z after death e
      nl[i] = fmax(pm, nl[i]);           // lmax after death e
      const double dz = logaddexp_dev(0.0, nl[i] + lvv[i] - lw[i]);
      // the reference tests at the top of the NEXT iteration, with the state this death leaves behind:
      // delta ln Z < dlogz, the last dead point above logl_max, its loop counter it = it0 + e + 1 > maxiter,
      // the calls spent so far > maxcall (sampler.py:1070-1093)
      bool stop = dz < a.dlogz || dcur[e] > a.logl_max || (a.maxiter >= 0 && it0 + e + 1 > a.maxiter);
      if (a.maxcall >= 0 && !stop) {  // (rarely asked for: a plain sum over the entries popped up to this death)
        long long cum = ncall0;
        for (int j = 0; j <= dj[e]; ++j) cum += qc[j];
        stop = cum > a.maxcall;
      }
      if (stop && e < mystop) mystop = e;
    }
  }
  if (mystop != 0x7fffffff) atomicMin(&misc[3], mystop);
  __syncthreads();
  const int estop = misc[3];
  const bool stopped = estop != 0x7fffffff;
  const int nkeep = stopped ? estop + 1 : ndead;
  const int E = nkeep - 1;
#pragma unroll
  for (int i = 0; i < kEPT; ++i)
    if (i < EPT && t * EPT + i == E) {
      bcast[0] = lw[i];
      bcast[1] = nl[i];
      bcast[3] = lvv[i];
      // the plateau mode after death E: what its plateau still has to take
      int pcn = 0;
      double plogn = 0.0;
      if (has_tie) {
        if (E < pc_in) {
          pcn = pc_in - nkeep;
          plogn = r.plogdvol;
        } else {
          int j, m;
          run_info(E, j, m);
          if (m > j) {
            pcn = m - j;
            // ln of the plateau's volume step: ln X at its start (ln X now minus the j steps taken) - ln(N + 1)
            plogn = lvv[i] - log1p(-(double)j / ((double)N + 1.0)) - log((double)N + 1.0);
          }
        }
      }
      pl_int[0] = pcn;
      pl_dbl[0] = plogn;
    }
  __syncthreads();
  const double logz_E = E >= 0 ? bcast[0] : logz0, lmax_E = E >= 0 ? bcast[1] : lmax0;
  NS_PROF(20);
  // information: sum of the L e^L dX terms relative to e^{lnZ_E}
  double hs = 0.0;
  double tt[kEPT];
  long long calls = 0;
  const int jlast = stopped ? dj[E] : (jcap >= 0 ? jcap : K - 1);
#pragma unroll
  for (int i = 0; i < kEPT; ++i) {
    const int e = t * EPT + i;
    tt[i] = 0.0;
    if (i < EPT && e <= E) {
      const double lnew = dcur[e], lprev = e ? dcur[e - 1] : dead_prev0;
      const double ldv = lvv[i] + (has_tie ? log(0.5 * expm1(cd[i])) : ldv_c);
      const double t0 = exp(lprev - logz_E + ldv), t1 = exp(lnew - logz_E + ldv);
      tt[i] = (t0 > 0.0 ? t0 * lprev : 0.0) + (t1 > 0.0 ? t1 * lnew : 0.0);
      hs += tt[i];
    }
  }
  // plateau deaths: their d ln X differs from the constant step, so their share of var[ln Z] = sum dH dlnX does not
  // telescope -- each needs its own dH = H_e - H_{e-1}, from the prefix sums of the terms above:
  //   H_e = e^{lnZ_E - lnZ_e} (w0 (H_0 + lnZ_0) + P_e) - lnZ_e,   P_e = sum_{k <= e} terms_k
  // -- and the two sums ns_finish needs for the same purpose (relative to Z_E): A += terms_e delta_e, B += (dZ_e / Z_E) delta_e
  double corr = 0.0, va_add = 0.0, vb_add = 0.0;
  if (has_tie) {
    const double ex = excl_sum(hs, sred[1]);
    const double w0 = exp(logz0 - logz_E);
    const double g0 = w0 > 0.0 ? w0 * (h0 + logz0) : 0.0;
    double pbefore = ex, lzbefore = pz;
#pragma unroll
    for (int i = 0; i < kEPT; ++i) {
      const int e = t * EPT + i;
      if (i < EPT && e <= E) {
        const double delta = cd[i] - dlv;
        if (delta != 0.0) {
          const double hprev = e == 0 ? h0 : exp(logz_E - lzbefore) * (g0 + pbefore) - lzbefore;
          const double he = exp(logz_E - lw[i]) * (g0 + pbefore + tt[i]) - lw[i];
          corr += (he - hprev) * delta;
          va_add += tt[i] * delta;
          vb_add += (exp(lw[i] - logz_E) - exp(lzbefore - logz_E)) * delta;
        }
        pbefore += tt[i];
        lzbefore = lw[i];
      }
    }
  }
  NS_PROF(21);
  for (int j = t; j <= jlast; j += kT) calls += qc[j];
  for (int off = 32; off > 0; off >>= 1) {
    hs += __shfl_xor(hs, off);
    calls += __shfl_xor(calls, off);
    corr += __shfl_xor(corr, off);
    va_add += __shfl_xor(va_add, off);
    vb_add += __shfl_xor(vb_add, off);
  }
  __syncthreads();  // (sred[1] / wred were read above)
  if (lane == 0) {
    wred[0][wv] = hs;
    lred[wv] = calls;
    sred[0][wv] = corr;
    sred[1][wv] = va_add;
    sred[2][wv] = vb_add;
  }
  NS_PROF(2);
  // ---- replay the walk up to the stop index (once per run): the sorted order is untouched ----
  if (nkeep < ndead) {
    __syncthreads();
    for (int i = t; i < NC; i += kT) src[i] = -1;
    __syncthreads();
    if (t < 64) {
      int jc;
      double nm;
      consume_sorted(skey, sidx, src, ql, dcur, dj, dslot, dsrc, bkey, bslot, bsrc, NC, K,
                     a.dead_rel ? (long long)K + 1 : a.cap - it0, nkeep, &jc, &nm, t);
      if (t == 0) bcast[2] = nm;
    }
  }
  __syncthreads();
  if (t == 0) {
    const double hsum = wred[0][0] + wred[0][1] + wred[0][2] + wred[0][3];
    r.ncall += lred[0] + lred[1] + lred[2] + lred[3];
    if (nkeep > 0) {
      const double w = exp(logz0 - logz_E);
      const double h_E = hsum + (w > 0.0 ? w * (h0 + logz0) : 0.0) - logz_E;
      r.logzvar += (h_E - h0) * dlv;
      r.h = h_E;
      r.logz = logz_E;
      r.lmax = lmax_E;
      r.dead_prev = dcur[E];
      r.logvol = has_tie ? bcast[3] : logvol0 - (double)nkeep * dlv;
      r.it = it0 + nkeep;
      if (has_tie || r.var_a != 0.0 || r.var_b != 0.0) {
        // (the plateau sums are kept relative to the running Z)
        r.logzvar += sred[0][0] + sred[0][1] + sred[0][2] + sred[0][3];
        r.var_a = (w > 0.0 ? r.var_a * w : 0.0) + (sred[1][0] + sred[1][1] + sred[1][2] + sred[1][3]);
        r.var_b = (w > 0.0 ? r.var_b * w : 0.0) + (sred[2][0] + sred[2][1] + sred[2][2] + sred[2][3]);
      }
      if (has_tie) {
        r.pcount = pl_int[0];
        r.plogdvol = pl_dbl[0];
      }
    }
    r.loglstar = bcast[2];
    r.nfill += 1;
    {
      // entries popped after the last kept death found no worse point to replace any more in this fill: the
      // reference goes on popping from the refilled queue within the same iteration, so their calls belong
      // to the next death's 'nc'
      long long c = nkeep > 0 ? 0 : carry0;
      for (int j = nkeep > 0 ? dj[E] + 1 : 0; j <= jlast; ++j) c += qc[j];
      r.nc_carry = c;
    }
    if (mode == MODE_BOUND && a.sampler != 3) {
      const int ta = racc[0], tr = rrej[0];
      if (a.sampler == 0) {
        // RWalkSampler.tune (internal_samplers.py:460-493), once per queue fill
        if (ta + tr > 0) r.scale *= exp(((double)ta / (double)(ta + tr) - a.facc) / (double)D / a.facc);
      } else {
        // tune_slice (internal_samplers.py:1209-1239)
        const double ne = ta > 1 ? (double)ta : 1.0, nt = (double)tr;
        double mult = ne * 2.0 / (ne + nt);
        mult = fmin(fmax(mult, 0.5), 2.0);
        r.scale *= mult;
      }
    }
    int done = stopped ? 1 : ((jcap >= 0 || atomicOr(&r.sampler_failed, 0)) ? 2 : 0);
    // The plateau mode's companion stop (sampler.py:1095-1100: np.ptp(live_logl) == 0 -> "we have reached the plateau
    // in the likelihood", the run ends normally): lmax is the largest value ever inserted and the point that carries
    // it is alive until it is the worst, so the live set has no spread exactly when its worst value has reached
    // lmax.  No proposal can beat such a threshold: without this stop the run would idle to max_fills (and the
    // uniform samplers retry 2^32 times per walker).  (The reference tests before every death, here once per fill.)
    if (!done && bcast[2] >= r.lmax) done = 1;
    if (done) {
      r.mode = done == 1 ? MODE_DONE : MODE_FAILED;
      atomicAdd(a.ndone, 1);
    }
  }
  NS_PROF(3);
  // dead-point log-likelihoods, in death order
  const size_t dbase = a.dead_rel ? (size_t)run * K : (size_t)run * a.cap + it0;
  for (int e = t; e < nkeep; e += kT) a.dead_logl[dbase + e] = dcur[e];
  if (a.trace_slot)
    for (int e = t; e < nkeep; e += kT) {
      a.trace_slot[(size_t)run * K + e] = real_slot(dslot[e]);
      a.trace_src[(size_t)run * K + e] = dj[e];
    }
  if (a.trace_n && t == 0) {
    a.trace_n[run * 2] = nkeep;
    a.trace_n[run * 2 + 1] = stopped ? 1 : 0;
  }
  if (a.dead_id) {
    for (int e = t; e < nkeep; e += kT) qborn[dj[e]] = e;
    __syncthreads();
    for (int e = t; e < nkeep; e += kT) {
      long long nc = e ? 0 : carry0;
      for (int j = e ? dj[e - 1] + 1 : 0; j <= dj[e]; ++j) nc += qc[j];
      const int sj = dsrc[e];
      a.dead_nc[dbase + e] = (int)nc;
      a.dead_id[dbase + e] = real_slot(dslot[e]);
      // self.it starts at 1 (sampler.py:396): the replacement of death index i (0-based) is born at i + 1
      a.dead_it[dbase + e] = sj < 0 ? a.live_it[(size_t)run * N + real_slot(dslot[e])] : (int)(it0 + qborn[sj] + 1);
    }
    __syncthreads();
    for (int sl = t; sl < NC; sl += kT)
      if (src[sl] >= 0) a.live_it[(size_t)run * N + real_slot(sl)] = (int)(it0 + qborn[src[sl]] + 1);
  }
  // dead-point coordinates (optional), in death order
  if (a.store_samples) {
    double* to = a.dead_u + ((size_t)run * a.cap + it0) * D;
    for (int x = t; x < nkeep * D; x += kT) {
      const int e = x / D, j = x - e * D;
      const int s = real_slot(dslot[e]), sj = dsrc[e];
      to[x] = sj < 0 ? a.live_u[((size_t)run * N + s) * D + j] : a.r_u[((size_t)run * K + sj) * D + j];
    }
  }
  if (a.undo_slot) {
    const int E1 = nkeep - 1;
    if (nkeep > 0 && dj[E1] == K - 1) {
      const int s = real_slot(dslot[E1]), sj = dsrc[E1];
      for (int x = t; x < D; x += kT)
        a.undo_u[(size_t)run * D + x] = sj < 0 ? a.live_u[((size_t)run * N + s) * D + x] : a.r_u[((size_t)run * K + sj) * D + x];
      if (t == 0) a.undo_slot[run] = s;
    } else if (t == 0) {
      a.undo_slot[run] = -1;
    }
  }
  __syncthreads();
  NS_PROF(4);
  // apply the surviving replacements to the live set: the replaced slots are listed first (dslot / dj are free by
  // now), then their rows are copied element-parallel (one thread per slot copying 2 D strided doubles was 1 ms of
  // the 1.7 ms this kernel took per fill at D = 200)
  if (t == 0) misc[4] = 0;
  __syncthreads();
  for (int s = t; s < NC; s += kT) {
    const int sj = src[s];
    if (sj >= 0) {
      a.live_logl[(size_t)run * N + real_slot(s)] = ql[sj];
      const int c = atomicAdd(&misc[4], 1);  // at most K slots change
      dslot[c] = real_slot(s);
      dj[c] = sj;
    }
  }
  __syncthreads();
  {
    const int nrep = misc[4];
    double* lu = a.live_u + (size_t)run * N * D;
    double* lv = a.live_v + (size_t)run * N * D;
    const double* ru = a.r_u + (size_t)run * K * D;
    const double* rv = a.r_v + (size_t)run * K * D;
    // (sixteen elements per thread in flight -- one at a time the loop ran at one global round trip per element, with
    // eight it was still five dependent trips per C2 fill: 20 k of the kernel's 108 k cycles)
    const int tot = nrep * D;
    constexpr int LS = 16;
    for (int e0 = t; e0 < tot; e0 += LS * kT) {
      size_t to[LS];
      double xu[LS], xv[LS];
#pragma unroll
      for (int q = 0; q < LS; ++q) {
        const int e = e0 + q * kT;
        const int ec = e < tot ? e : 0;
        const int c = ec / D, j = ec - c * D;
        to[q] = (size_t)dslot[c] * D + j;
        const size_t from = (size_t)dj[c] * D + j;
        xu[q] = ru[from];
        xv[q] = rv[from];
      }
#pragma unroll
      for (int q = 0; q < LS; ++q)
        if (e0 + q * kT < tot) {
          lu[to[q]] = xu[q];
          lv[to[q]] = xv[q];
        }
    }
  }
  NS_PROF(5);
}

// ---- final live points (sampler.py:780-930) + record -----------------------------
// exclusive prefix of one value per thread over the workgroup (threads in order), and the total; OP = logaddexp or +
template <bool LOGADD>
__device__ __forceinline__ double block_excl_scan(double v, double ident, double* wtot /* 4 shared */, double* total) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  double sc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const double y = __shfl_up(sc, off);
    if (lane >= off) sc = LOGADD ? logaddexp_dev(y, sc) : y + sc;
  }
  double ex = __shfl_up(sc, 1);
  if (lane == 0) ex = ident;
  __syncthreads();
  if (lane == 63) wtot[wv] = sc;
  __syncthreads();
  double pre = ident, tot = ident;
  for (int w2 = 0; w2 < kT / 64; ++w2) {
    if (w2 < wv) pre = LOGADD ? logaddexp_dev(pre, wtot[w2]) : pre + wtot[w2];
    tot = LOGADD ? logaddexp_dev(tot, wtot[w2]) : tot + wtot[w2];
  }
  *total = tot;
  return LOGADD ? logaddexp_dev(pre, ex) : pre + ex;
}

__global__ void __launch_bounds__(kT) ns_finish(NsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int run = blockIdx.x, t = threadIdx.x, N = a.nlive;
  if (!a.add_live) {
    // run_nested(add_live=False): the results are the loop's running integrals (progress_integration's values after
    // the last death), the live points stay out (sampler.py:1319-1341)
    if (t == 0) {
      const NsRun& r = a.st[run];
      double* rec = a.records + (size_t)run * 8;
      rec[0] = r.logz;
      rec[1] = sqrt(fabs(r.logzvar));
      rec[2] = (double)r.it;
      rec[3] = (double)r.ncall;
      rec[4] = r.h;
      rec[5] = (double)r.nbound;
      rec[6] = (double)(r.mode == MODE_DONE ? 0 : (r.mode == MODE_FAILED ? -1 : 1));
      rec[7] = 100.0 * (double)r.it / (double)r.ncall;
    }
    return;
  }
  // the final live log-likelihoods, ascending: the slots sorted by (value, slot) with the register network of
  // ns_consume, then the values in that order (the integration below does not care which slot a value came from).
  // Live sets beyond the sort's reach or the LDS (ns_finish_big): the values alone, by a bitonic network in global
  // memory behind the per-point terms (once per run; padded with +inf to a power of two).
  int P = 1;
  while (P < N) P <<= 1;
  const int nsidx = P > kT ? P : kT;
  double* sorted;
  if (ns_finish_big(N)) {
    sorted = a.fin_ws + (size_t)run * a.fin_stride + 3 * (size_t)N;
    for (int i = t; i < P; i += kT) sorted[i] = i < N ? a.live_logl[(size_t)run * N + i] + 0.0 : INFINITY;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
      for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
        for (int x = t; x < P / 2; x += kT) {
          const int lo = 2 * x - (x & (j2 - 1)), hi = lo + j2;  // the pair (lo, lo ^ j2) with lo's bit j2 clear
          const double va = sorted[lo], vb = sorted[hi];
          const bool up = (lo & k2) == 0;
          if ((va > vb) == up) {
            sorted[lo] = vb;
            sorted[hi] = va;
          }
        }
        __syncthreads();
      }
  } else {
    double* skey = (double*)smem;                                // N   by slot
    sorted = skey + N;                                           // N   ascending
    unsigned short* sidx = (unsigned short*)(sorted + N);        // max(P, kT)
    for (int i = t; i < N; i += kT) skey[i] = a.live_logl[(size_t)run * N + i];
    __syncthreads();
    switch (nsidx / kT) {
      case 1: sort_slots<1>(skey, sidx, N, nsidx); break;
      case 2: sort_slots<2>(skey, sidx, N, nsidx); break;
      case 4: sort_slots<4>(skey, sidx, N, nsidx); break;
      case 8: sort_slots<8>(skey, sidx, N, nsidx); break;
      case 16: sort_slots<16>(skey, sidx, N, nsidx); break;
      default: sort_slots<32>(skey, sidx, N, nsidx); break;
    }
    for (int i = t; i < N; i += kT) sorted[i] = skey[sidx[i]];
    __syncthreads();
  }
  // The final live points, lowest first (sampler.py:780-930).  What Results reports is
  // compute_integrals over the whole run (sampler.py:1342-1348, utils.py:1411-1467): ln Z the
  // accumulated logaddexp, and the partial informations H_i normalised by the FINAL Z,
  //   H_i = (1/Z_f) sum_{k<=i} [L ln L]-terms - (Z_i / Z_f) ln Z_f,   var ln Z = |sum_i (H_i - H_{i-1}) dlnX_i|.
  // With G = e^{lnZ}(H + lnZ) (additive) the state after the dead points gives
  // H_n = e^{lnZ_n - lnZ_f} (H^run_n + lnZ_n - lnZ_f); over the dead points dlnX is constant, so their
  // share of the variance sum telescopes to dlnX * H_n.
  // Every thread owns a contiguous stretch of the points; the two running quantities -- ln Z_i (logaddexp) and the sum
  // of the L e^L terms -- are prefix scans over the stretches' totals, after which a thread walks its own stretch
  // (one thread walking all N points in the reference's order was 1.4 of this kernel's 6.5 ms at N = 5 000; the
  // results agree to 1e-13).
  __shared__ double wtot[kT / 64];
  NsRun& r = a.st[run];
  const double lv0 = r.logvol, dead_prev = r.dead_prev;
  double* W = a.fin_ws + (size_t)run * a.fin_stride;  // trapezoid ln-weights
  double* DL = W + N;                          // -d ln X
  double* T = DL + N;                          // ln dX, then the L e^L terms
  // a plateau still being worked off keeps its volume steps for its remaining points, the rest of the final
  // points share what is left (sampler.py:813-830)
  const int pc = r.pcount < N ? r.pcount : N;
  const double pstep = pc > 0 ? exp(r.plogdvol - lv0) : 0.0;
  auto lv_at = [&](int i) {  // ln X after the i-th final point (1-based; 0: before the first)
    if (pc == 0) return i > 0 ? lv0 + log(1.0 - (double)i / ((double)N + 1.0)) : lv0;
    if (i <= pc) return lv0 + log1p(-(double)i * pstep);
    return lv0 + log1p(-(double)pc * pstep) + log1p(-(double)(i - pc) / ((double)(N - pc) + 1.0));
  };
  const int per = (N + kT - 1) / kT, c0 = t * per < N ? t * per : N, c1 = c0 + per < N ? c0 + per : N;
  double zc = -INFINITY;  // ln of the stretch's summed weights
  for (int i = c0 + 1; i <= c1; ++i) {
    const double cur = sorted[i - 1], prev = i > 1 ? sorted[i - 2] : dead_prev;
    const double lv = lv_at(i);
    const double lvprev = lv_at(i - 1);
    const double dl = lvprev - lv;
    const double logdvol = lv + log(0.5 * expm1(dl));
    const double w = logaddexp_dev(cur, prev) + logdvol;
    W[i - 1] = w;
    DL[i - 1] = dl;
    T[i - 1] = logdvol;
    zc = logaddexp_dev(zc, w);
  }
  double ztot;
  double zpre = block_excl_scan<true>(zc, -INFINITY, wtot, &ztot);
  const double logz_f = logaddexp_dev(r.logz, ztot);
  zpre = logaddexp_dev(r.logz, zpre);  // ln Z before this thread's stretch
  double hc = 0.0;
  for (int i = c0 + 1; i <= c1; ++i) {
    const double cur = sorted[i - 1], prev = i > 1 ? sorted[i - 2] : dead_prev;
    const double logdvol = T[i - 1];
    const double t0 = exp(prev - logz_f + logdvol), t1 = exp(cur - logz_f + logdvol);
    const double tt = (t1 > 0.0 ? t1 * cur : 0.0) + (t0 > 0.0 ? t0 * prev : 0.0);
    T[i - 1] = tt;
    hc += tt;
  }
  double htot;
  const double hpre = block_excl_scan<false>(hc, 0.0, wtot, &htot);
  const double dlv = log(((double)N + 1.0) / (double)N);
  const double wn = exp(r.logz - logz_f);
  const double hpart0 = r.it > 0 && wn > 0.0 ? wn * (r.h + r.logz) : 0.0;  // (1/Z_f) sum of the L ln L terms so far
  // H before this thread's stretch (for the first stretch: H_n)
  double logz = zpre, hpart = hpart0 + hpre;
  double hcur = hpart - (c0 == 0 ? (r.it > 0 ? wn * logz_f : 0.0) : logz_f * exp(logz - logz_f));
  double var = 0.0;
  for (int i = c0; i < c1; ++i) {
    logz = logaddexp_dev(logz, W[i]);
    hpart += T[i];
    const double hi = hpart - logz_f * exp(logz - logz_f);
    var += (hi - hcur) * DL[i];
    hcur = hi;
  }
  __shared__ double fin_h;
  if (c1 == N && c0 < N) fin_h = hcur;  // the thread that owns the last point
  double vtot;
  (void)block_excl_scan<false>(var, 0.0, wtot, &vtot);
  __syncthreads();
  if (t == 0) {
    // the dead points' share of sum_i (H_i - H_{i-1}) dlnX_i: it telescopes for the constant step, and the
    // plateau deaths' departures from it were summed on the way (ns_consume: var_a, var_b, relative to Z_n)
    const double h_n = hpart0 - (r.it > 0 ? wn * logz_f : 0.0);
    const double logzvar = h_n * dlv + (wn > 0.0 ? wn * (r.var_a - logz_f * r.var_b) : 0.0) + vtot;
    double* rec = a.records + (size_t)run * 8;
    rec[0] = logz_f;
    rec[1] = sqrt(fabs(logzvar));
    rec[2] = (double)r.it;
    rec[3] = (double)r.ncall;
    rec[4] = N > 0 ? fin_h : h_n;
    rec[5] = (double)r.nbound;
    rec[6] = (double)(r.mode == MODE_DONE ? 0 : (r.mode == MODE_FAILED ? -1 : 1));  // 1: hit maxfills
    rec[7] = 100.0 * (double)r.it / (double)r.ncall;
  }
}

}  // namespace

namespace {
// the instance of ns_consume for a queue of K entries (kEPT = 1, 2, 4 or 8 deaths per lane)
typedef void (*NsConsumeFn)(NsArgs);
inline NsConsumeFn ns_consume_for(int K) {
  const int ept = (K + kT - 1) / kT;
  return ept <= 1 ? ns_consume<1> : ept <= 2 ? ns_consume<2> : ept <= 4 ? ns_consume<4> : ns_consume<8>;
}
}  // namespace

extern "C" {

// see include/dynhip.h
int dh_ns_consume(dh_ctx* ctx, int runs, int nlive, int queue_size, double dlogz, double* live_logl,
                  const double* q_logl, const int32_t* q_ncalls, double* state, double* dead_logl,
                  int32_t* dead_slot, int32_t* dead_src, int32_t* ndead, int32_t* stopped, int32_t* live_it,
                  int32_t* dead_it, int32_t* dead_nc, double* plateau) {
  DH_CHECK_CTX(ctx);
  const bool want_pt = live_it || dead_it || dead_nc;
  if (want_pt && !(live_it && dead_it && dead_nc))
    return fail(ctx, DH_ERR_ARG, "ns_consume: live_it, dead_it and dead_nc come together");
  if (runs < 1 || nlive < 4 || queue_size < 1 || !live_logl || !q_logl || !q_ncalls || !state || !dead_logl ||
      !dead_slot || !dead_src || !ndead || !stopped)
    return fail(ctx, DH_ERR_ARG, "ns_consume: bad arguments");
  const int R = runs, N = nlive, K = queue_size;
  if (K > kEPTMax * kT) return fail(ctx, DH_ERR_ARG, "ns_consume: queue_size %d > %d", K, kEPTMax * kT);
  // slots travel as 16-bit indices; beyond the register sort's 32 keys per thread (or the LDS) the consumption works on
  // the K + 1 smallest live points (ns_consume_compact)
  if (N > 65535) return fail(ctx, DH_ERR_ARG, "ns_consume: nlive %d > 65535", N);
  const size_t lds_max = ns_consume_lds(N, K);
  if (lds_max > 150 * 1024) return fail(ctx, DH_ERR_ARG, "ns_consume: nlive/queue too large for LDS");
  NsArgs a{};
  a.serial_walk = (getenv("DH_NS_SERIAL") && atoi(getenv("DH_NS_SERIAL")) != 0) ? 1 : 0;
  a.runs = R;
  a.nlive = N;
  a.ndim = 0;  // log-likelihoods only: no coordinates travel
  a.K = K;
  a.walks = 1;
  a.cap = K;
  a.maxiter = a.maxcall = -1;
  a.logl_max = INFINITY;
  a.add_live = 1;
  a.forced_exact = 0;
  a.force_first = nullptr;
  a.presort = nullptr;
  a.presort_stride = 0;
  a.presorted = 0;
  a.sel_ent = nullptr;
  a.undo_u = nullptr;
  a.undo_slot = nullptr;
  a.dlogz = dlogz;
  a.dead_rel = 1;
  arena_reset(ctx);
  int rc = arena_reserve(ctx, (size_t)R * (sizeof(NsRun) + (size_t)N * 24 + (size_t)K * 40 + 64) + 16384);
  if (rc) return rc;
  std::vector<NsRun> st((size_t)R);
  for (int r = 0; r < R; ++r) {
    NsRun& x = st[(size_t)r];
    memset(&x, 0, sizeof x);
    const double* sv = state + (size_t)r * 8;
    x.logvol = sv[0];
    x.logz = sv[1];
    x.h = sv[2];
    x.logzvar = sv[3];
    x.dead_prev = sv[4];
    x.it = (long long)sv[5];
    x.ncall = (long long)sv[6];
    x.mode = MODE_CUBE;  // the queue's ncalls come from q_ncalls; no sampler tuning
    x.scale = 1.0;
    if (plateau) {  // plateau mode carried from the previous call (sampler.py:1112-1127)
      x.pcount = (int)plateau[(size_t)r * 2];
      x.plogdvol = plateau[(size_t)r * 2 + 1];
    }
  }
  a.st = arena_up(ctx, st.data(), (size_t)R);
  a.live_logl = arena_up(ctx, live_logl, (size_t)R * N);
  a.dead_logl = (double*)arena_get(ctx, (size_t)R * K * 8);
  a.r_logl = arena_up(ctx, q_logl, (size_t)R * K);
  a.r_a = (int*)arena_up(ctx, q_ncalls, (size_t)R * K);
  a.trace_slot = (int*)arena_get(ctx, (size_t)R * K * 4);
  a.trace_src = (int*)arena_get(ctx, (size_t)R * K * 4);
  a.trace_n = (int*)arena_get(ctx, (size_t)R * 8);
  a.ndone = (int*)arena_get(ctx, 64);
  a.bstatus = (int*)arena_get(ctx, (size_t)R * 4);
  if (want_pt) {
    a.live_it = (int*)arena_up(ctx, (const int*)live_it, (size_t)R * N);
    a.dead_id = (int*)arena_get(ctx, (size_t)R * K * 4);
    a.dead_it = (int*)arena_get(ctx, (size_t)R * K * 4);
    a.dead_nc = (int*)arena_get(ctx, (size_t)R * K * 4);
    if (!a.live_it || !a.dead_id || !a.dead_it || !a.dead_nc) return DH_ERR_NOMEM;
  }
  if (!a.st || !a.live_logl || !a.dead_logl || !a.r_logl || !a.r_a ||
      !a.trace_slot || !a.trace_src || !a.trace_n || !a.ndone || !a.bstatus)
    return DH_ERR_NOMEM;
  hipStream_t s = ctx->stream;
  if (!hip_ok(ctx, hipMemsetAsync(a.ndone, 0, 64, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(a.bstatus, 0, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(a.trace_n, 0, (size_t)R * 8, s), "memset"))
    return DH_ERR_HIP;
  if (!hip_ok(ctx, hipFuncSetAttribute((const void*)ns_consume_for(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), "hipFuncSetAttribute(ns_consume)") ||
      !hip_ok(ctx, hipFuncSetAttribute((const void*)ns_start, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), "hipFuncSetAttribute(ns_start)"))
    return DH_ERR_HIP;
  hipLaunchKernelGGL(ns_start, dim3(R), dim3(kT), 0, s, a);  // loglstar = min, lmax = max of live_logl
  hipLaunchKernelGGL(ns_consume_for(K), dim3(R), dim3(kT), lds_max, s, a);
  if (!hip_ok(ctx, hipGetLastError(), "ns_consume launch")) return DH_ERR_HIP;
  std::vector<int> tn((size_t)R * 2);
  if (!down(ctx, st.data(), a.st, (size_t)R) || !down(ctx, live_logl, a.live_logl, (size_t)R * N) ||
      !down(ctx, dead_logl, a.dead_logl, (size_t)R * K) || !down(ctx, (int*)dead_slot, a.trace_slot, (size_t)R * K) ||
      !down(ctx, (int*)dead_src, a.trace_src, (size_t)R * K) || !down(ctx, tn.data(), a.trace_n, (size_t)R * 2))
    return DH_ERR_HIP;
  if (want_pt && (!down(ctx, (int*)live_it, a.live_it, (size_t)R * N) || !down(ctx, (int*)dead_it, a.dead_it, (size_t)R * K) ||
                  !down(ctx, (int*)dead_nc, a.dead_nc, (size_t)R * K)))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  for (int r = 0; r < R; ++r) {
    const NsRun& x = st[(size_t)r];
    double* sv = state + (size_t)r * 8;
    sv[0] = x.logvol;
    sv[1] = x.logz;
    sv[2] = x.h;
    sv[3] = x.logzvar;
    sv[4] = x.dead_prev;
    sv[5] = (double)x.it;
    sv[6] = (double)x.ncall;
    sv[7] = x.loglstar;  // the current worst live point
    if (plateau) {
      plateau[(size_t)r * 2] = (double)x.pcount;
      plateau[(size_t)r * 2 + 1] = x.plogdvol;
    }
    ndead[r] = tn[(size_t)r * 2];
    stopped[r] = tn[(size_t)r * 2 + 1];
  }
  return DH_OK;
}

int dh_ns_set_option(dh_ctx* ctx, int key, double value) {
  DH_CHECK_CTX(ctx);
  if (key < 0 || key >= DH_NS_OPT_COUNT) return fail(ctx, DH_ERR_ARG, "ns option %d", key);
  ctx->ns_opt[key] = value;
  return DH_OK;
}

int dh_ns_set_boundary(dh_ctx* ctx, int ndim, const int8_t* bc) {
  DH_CHECK_CTX(ctx);
  // validate first: a rejected call leaves the installed flags as they were
  if (bc && ndim > 0) {
    for (int i = 0; i < ndim; ++i)
      if (bc[i] != DH_BC_HARD && bc[i] != DH_BC_PERIODIC && bc[i] != DH_BC_REFLECT)
        return fail(ctx, DH_ERR_ARG, "ns boundary flag %d of dimension %d", (int)bc[i], i);
    ctx->ns_bc.assign(bc, bc + ndim);
  } else {
    ctx->ns_bc.clear();
  }
  return DH_OK;
}

int dh_ns_ensemble(dh_ctx* ctx, int problem, int runs, int nlive, int ndim, int queue_size, int sampler,
                   int walks, int bound_multi, int rebuild_sync, double dlogz, double enlarge, int64_t max_fills,
                   int64_t max_iter,
                   const uint32_t* entropy_words, int n_words, uint32_t first_run, double* records,
                   double* dead_logl_out, double* live_logl_out, double* dead_u_out, double* live_u_out,
                   int64_t* n_fills_out, int32_t* dead_id_out, int32_t* dead_it_out, int32_t* dead_nc_out,
                   int32_t* live_it_out, int bootstrap, int rebuild_every) {
  DH_CHECK_CTX(ctx);
  // Options and boundary flags are one-shot: this call takes them and the context forgets them, whichever way the call
  // ends, so that nothing set for one ensemble (maxiter, add_live = 0, forced_exact, periodic coordinates) can leak
  // into a later one whose caller set nothing.
  struct NsOneShot {
    dh_ctx* c;
    double opt[DH_NS_OPT_COUNT];
    std::vector<int8_t> bc;
    explicit NsOneShot(dh_ctx* cc) : c(cc) {
      for (int i = 0; i < DH_NS_OPT_COUNT; ++i) {
        opt[i] = c->ns_opt[i];
        c->ns_opt[i] = __builtin_nan("");
      }
      bc.swap(c->ns_bc);
    }
  } shot(ctx);
  const bool want_pt = dead_id_out || dead_it_out || dead_nc_out || live_it_out;
  if (want_pt && !(dead_id_out && dead_it_out && dead_nc_out && live_it_out))
    return fail(ctx, DH_ERR_ARG, "ns_ensemble: the per-point outputs (id, it, nc, live it) come together");
  ProblemDev pd;
  if (!get_problem(ctx, problem, &pd)) return DH_ERR_ARG;
  if (pd.ndim != ndim) return fail(ctx, DH_ERR_ARG, "problem ndim %d != %d", pd.ndim, ndim);
  // sampler 3 / 4 / 5 = rwalk / rslice / slice with the unit-cube phase and the proposals drawn from hiprand
  // Philox streams (throughput RNG mode); start points and frames keep their PCG64 streams
  // sampler 6 / 7 = the uniform sampler inside the bound (UniformBoundSampler) from PCG64 / Philox streams
  const bool philox = (sampler >= 3 && sampler <= 5) || sampler == 7;
  if (sampler >= 6 && sampler <= 7)
    sampler = 3;  // internal code of `unif`
  else if (philox)
    sampler -= 3;
  else if (sampler > 2)
    sampler = -1;
  if (runs < 1 || nlive < 4 || queue_size < 1 || walks < 1 || sampler < 0 || sampler > 3 || !entropy_words ||
      n_words < 1 || !records || bootstrap < 0 || bootstrap == 1)
    return fail(ctx, DH_ERR_ARG, "ns_ensemble: bad arguments");
  // Above the register-resident dimensions (and for slice samplers at dimensions without an instantiation) the
  // walker launches go to the wave-per-walker kernels of wide.hip, which take the same per-run arrays; above
  // d = 44 the bound is the multi-workgroup Ellipsoid.update with the run mask, or -- bound='multi' -- the wide
  // MultiEllipsoid.update: a host recursion over device node work, so a rebuild fill there synchronises the stream
  // and reads the run mask back (the loop is no longer launch-ahead on those fills; the tree is a handful of nodes).
  const int N = nlive, D = ndim, K = queue_size, R = runs;
  // the register sort of the live slots is built for at most 32 keys per thread (sort_slots<32>); the LDS bound of
  // ns_finish below is tighter today, this one is the sort's own
  if (N > 65535) return fail(ctx, DH_ERR_ARG, "ns_ensemble: nlive %d > 65535 (slots travel as 16-bit indices)", N);
  if (!shot.bc.empty() && (int)shot.bc.size() != ndim)
    return fail(ctx, DH_ERR_ARG, "ns_ensemble: %d boundary flags for ndim %d", (int)shot.bc.size(), ndim);
  const int me = bound_multi ? (N / (2 * D) > 0 ? N / (2 * D) : 1) : 1;
  NsArgs a{};
  a.runs = R;
  a.nlive = N;
  a.ndim = D;
  a.K = K;
  a.walks = walks;
  a.sampler = sampler;
  a.bound_multi = bound_multi;
  a.max_ells = me;
  a.cap = max_iter > 0 ? max_iter : 400000;
  a.dlogz = dlogz;
  a.enlarge_log = log(enlarge);
  a.facc = fmin(1.0, fmax(1.0 / (double)(walks > 2 ? walks : 2), 0.5));
  a.first_eff = 10.0;
  a.first_ncall = 2ll * N;
  // update_bound_interval_ratio (internal_samplers.py:495-502, 581-588, 737-744) * nlive
  // (UniformBoundSampler keeps the base class's ratio 1, internal_samplers.py:88-94)
  a.update_interval = (long long)(sampler == 3 ? 1 : sampler == 2 ? walks * D : walks) * N;
  // the sampler's / run_nested's options a caller has set (dh_ns_set_option; NaN = the reference's default above)
  a.maxiter = a.maxcall = -1;
  a.logl_max = INFINITY;
  a.add_live = 1;
  a.forced_exact = 0;
  a.force_first = nullptr;
  a.presort = nullptr;
  a.presort_stride = 0;
  a.presorted = 0;
  a.sel_ent = nullptr;
  a.undo_u = nullptr;
  a.undo_slot = nullptr;
  {
    const double* o = shot.opt;
    if (!std::isnan(o[DH_NS_OPT_UPDATE_INTERVAL])) {
      // update_interval as dynesty takes it (dynesty.py:213-234): a float is a multiple of nlive, an int a number of calls
      const double v = o[DH_NS_OPT_UPDATE_INTERVAL];
      // (a value below one call is one call: the reference's max(min(round(...), maxsize), 1), dynesty.py:646-649)
      a.update_interval = v >= 1.0 ? (long long)llround(v) : 1;
    }
    if (!std::isnan(o[DH_NS_OPT_FIRST_MIN_NCALL])) a.first_ncall = (long long)llround(o[DH_NS_OPT_FIRST_MIN_NCALL]);
    if (!std::isnan(o[DH_NS_OPT_FIRST_MIN_EFF])) a.first_eff = o[DH_NS_OPT_FIRST_MIN_EFF];
    if (!std::isnan(o[DH_NS_OPT_MAXITER])) a.maxiter = (long long)llround(o[DH_NS_OPT_MAXITER]);
    if (!std::isnan(o[DH_NS_OPT_MAXCALL])) a.maxcall = (long long)llround(o[DH_NS_OPT_MAXCALL]);
    if (!std::isnan(o[DH_NS_OPT_LOGL_MAX])) a.logl_max = o[DH_NS_OPT_LOGL_MAX];
    if (!std::isnan(o[DH_NS_OPT_ADD_LIVE])) a.add_live = o[DH_NS_OPT_ADD_LIVE] != 0.0 ? 1 : 0;
    // default (round 5): the reference's protocol; 0 = the late form (opt-in fast mode)
    // (the uniform sampler has no start points, hence no forced update: for it the option is the ordering of the
    // regular update alone)
    a.forced_exact = 1;
    if (!std::isnan(o[DH_NS_OPT_FORCED_EXACT])) a.forced_exact = o[DH_NS_OPT_FORCED_EXACT] != 0.0 ? 1 : 0;
    if (a.maxcall >= 0 && a.maxcall < N)
      return fail(ctx, DH_ERR_ARG, "ns_ensemble: maxcall %lld below the %d calls of the initial live points", a.maxcall, N);
  }
  a.bootstrap = bootstrap;
  a.store_samples = dead_u_out ? 1 : 0;
  a.rebuild_sync = rebuild_sync ? 1 : 0;
  a.serial_walk = (getenv("DH_NS_SERIAL") && atoi(getenv("DH_NS_SERIAL")) != 0) ? 1 : 0;
  // Runs are independent, so WHEN a run's bound is rebuilt relative to the other runs' walks is free: a run that is
  // due sits the fill out (run_mode MODE_WAIT) while its rebuild -- a latency chain that leaves most of the chip idle
  // -- runs on a second stream beside the other runs' walkers, and walks from the new bound in the next fill.  Its own
  // sequence (rebuild, then walk from the same live set with the same generator state) is unchanged: with PCG64
  // streams every run's result is bit-identical to the serial schedule's (tests).  MEASURED, AND OFF BY DEFAULT
  // (DH_NS_OVERLAP=1 switches it on): a rebuild beside a walk slows both (77 KB of LDS and 256 VGPRs per rebuild
  // workgroup against two 230-VGPR walk workgroups per CU), and every run spends one more fill per bound update --
  // 64 C2 runs 0.336 -> 0.328 s, 16 eggbox runs 0.127 -> 0.160 s, 16 C4 runs 11.1 -> 11.3 s.
  const bool want_overlap = getenv("DH_NS_OVERLAP") && atoi(getenv("DH_NS_OVERLAP")) != 0;
  if (want_overlap && a.forced_exact) {
    // (the reference's protocol builds bounds inside the fill that needs them: nothing to overlap.  Said once, not
    // silently ignored -- the protocol is the default since round 5, also for C callers that never set the option)
    static bool warned = false;
    if (!warned) {
      fprintf(stderr, "dynhip: DH_NS_OVERLAP=1 has no effect while DH_NS_OPT_FORCED_EXACT is on (the default); "
                      "set the option to 0 for the late form\n");
      warned = true;
    }
  }
  a.overlap = (want_overlap && !a.forced_exact) ? 1 : 0;
  // Bounds are built every rebuild_every-th fill (runs that become due in between wait, see ns_prepare); 0 = chosen
  // here.  Once the period reaches the number of fills a run needs to spend its update interval, EVERY run is due (and
  // waiting) by the next rebuild fill: the ensemble rebuilds together and walks together, one latency chain per
  // interval instead of one per fill, with each run's own sequence untouched.  Longer periods only add idle fills
  // (a fill in which every run waits costs its launches, ~75 us), shorter ones lose the synchrony -- measured flat
  // from the interval upwards (64 C2 runs: 0.335 s at 1, 0.246 at 2, 0.29 at 3, 0.20 at 4 ... 12).  The interval in
  // fills is nlive / K for rwalk (every walker spends `walks` calls), about nlive / (4.4 K) for the slice samplers
  // (4.4 evaluations per slice step measured) and nlive / (1.7 K) for the uniform sampler; rounded up generously.
  // The choice is a function of the arguments only -- never of timings -- so a run's global fill indices, and with
  // them its Philox offsets, are reproducible.
  int every = rebuild_every;
  if (const char* e = getenv("DH_NS_REBUILD_EVERY")) every = atoi(e);
  if (every <= 0) {
    const double I = sampler == 0 ? (double)N / K : 1.3 * (double)N / ((sampler == 3 ? 1.7 : 4.4) * K);
    every = (int)ceil(I - 1e-9);
  }
  if (every < 1) every = 1;
  if (every > 16) every = 16;
  if (a.overlap) every = 1;
  a.prof = nullptr;
  if (getenv("DH_NS_PROF")) {
    if (hipMalloc((void**)&a.prof, 32 * sizeof(long long)) != hipSuccess) a.prof = nullptr;
    if (a.prof) (void)hipMemset(a.prof, 0, 32 * sizeof(long long));
  }
  // ---- one allocation for all state ----
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t dd = (size_t)D * D;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += al(bytes);
    return o;
  };
  const size_t o_st = take(sizeof(NsRun) * R), o_lu = take((size_t)R * N * D * 8), o_lv = take((size_t)R * N * D * 8),
               o_ll = take((size_t)R * N * 8),
               o_dl = take((size_t)R * a.cap * 8), o_du = take(dead_u_out ? (size_t)R * a.cap * D * 8 : 8), o_qu = take((size_t)R * K * D * 8),
               o_qf = take((size_t)R * K * 4), o_qr = take((size_t)R * K * 32), o_qo = take((size_t)R * K * 32),
               o_ru = take((size_t)R * K * D * 8), o_rv = take((size_t)R * K * D * 8),
               o_rl = take((size_t)R * K * 8), o_ra = take((size_t)R * K * 4), o_rb = take((size_t)R * K * 4),
               o_rc = take((size_t)R * K * 4), o_rd = take((size_t)R * K * 4), o_dbl = take((size_t)R * 4),
               o_pl = take((size_t)R * 8), o_ps = take((size_t)R * 8), o_pm = take((size_t)R * 4),
               o_rm = take((size_t)R * 4), o_fo = take((size_t)R * 4), o_nd = take(64), o_ne = take((size_t)R * 4),
               o_bs = take((size_t)R * 4), o_bc = take((size_t)R * me * D * 8), o_bv = take((size_t)R * me * dd * 8),
               o_ba = take((size_t)R * me * dd * 8), o_bx = take((size_t)R * me * dd * 8 * (a.forced_exact ? 2 : 1)),
               o_bl = take((size_t)R * me * D * 8), o_bg = take((size_t)R * me * 8),
               o_rec = take((size_t)R * 8 * 8), o_ent = take((size_t)n_words * 4),
               o_fw = take((size_t)R * ns_fin_stride(N) * 8), o_cum = take((size_t)R * me * 8), o_be = take((size_t)R * 32),
               o_rs = take((size_t)R * 8), o_ff = take((size_t)R * 4), o_se = take((size_t)R * 32),
               o_bcf = take((size_t)D + 8), o_uu = take((size_t)R * D * 8), o_us = take((size_t)R * 4),
               o_fk = take((size_t)R * 4), o_pmk = take((size_t)R * 4),
               o_boot = take(bootstrap > 0 ? bootstrap_ws_bytes(R, N, D, me, bootstrap) : 8),
               o_lit = take(want_pt ? (size_t)R * N * 4 : 8), o_pid = take(want_pt ? (size_t)R * a.cap * 4 : 8),
               o_pit = take(want_pt ? (size_t)R * a.cap * 4 : 8), o_pnc = take(want_pt ? (size_t)R * a.cap * 4 : 8),
               o_pso = take((size_t)R * 2048 * 2);
  char* base = nullptr;
  (void)hipSetDevice(ctx->device);
  if (!hip_ok(ctx, hipMalloc((void**)&base, off), "hipMalloc(ns state)")) return DH_ERR_NOMEM;
  hipStream_t main_stream = ctx->stream, rb_stream = nullptr;
  hipEvent_t ev_prep = nullptr, ev_rb = nullptr;
  auto cleanup = [&](int rc) {
    ctx->stream = main_stream;
    if (rb_stream) (void)hipStreamSynchronize(rb_stream);
    (void)hipStreamSynchronize(ctx->stream);
    if (ev_prep) (void)hipEventDestroy(ev_prep);
    if (ev_rb) (void)hipEventDestroy(ev_rb);
    if (rb_stream) (void)hipStreamDestroy(rb_stream);
    (void)hipFree(base);
    return rc;
  };
  if (a.overlap &&
      (!hip_ok(ctx, hipStreamCreateWithFlags(&rb_stream, hipStreamNonBlocking), "hipStreamCreate(rebuild)") ||
       !hip_ok(ctx, hipEventCreateWithFlags(&ev_prep, hipEventDisableTiming), "hipEventCreate") ||
       !hip_ok(ctx, hipEventCreateWithFlags(&ev_rb, hipEventDisableTiming), "hipEventCreate")))
    return cleanup(DH_ERR_HIP);
  a.st = (NsRun*)(base + o_st);
  a.live_u = (double*)(base + o_lu);
  a.live_v = (double*)(base + o_lv);
  a.live_logl = (double*)(base + o_ll);
  a.dead_logl = (double*)(base + o_dl);
  a.dead_u = dead_u_out ? (double*)(base + o_du) : nullptr;
  a.q_u0 = (double*)(base + o_qu);
  a.q_frame = (int*)(base + o_qf);
  a.q_rng = (uint64_t*)(base + o_qr);
  a.q_rng_out = (uint64_t*)(base + o_qo);
  a.r_u = (double*)(base + o_ru);
  a.r_v = (double*)(base + o_rv);
  a.r_logl = (double*)(base + o_rl);
  a.r_a = (int*)(base + o_ra);
  a.r_b = (int*)(base + o_rb);
  a.r_c = (int*)(base + o_rc);
  a.r_d = (int*)(base + o_rd);
  a.run_doubling = (int*)(base + o_dbl);
  a.run_loglstar = (double*)(base + o_pl);
  a.run_scale = (double*)(base + o_ps);
  a.run_mode = (int*)(base + o_pm);
  a.rebuild_mask = (int*)(base + o_rm);
  a.force = (int*)(base + o_fo);
  a.ndone = (int*)(base + o_nd);
  a.nells = (int*)(base + o_ne);
  a.bstatus = (int*)(base + o_bs);
  a.b_ctrs = (double*)(base + o_bc);
  a.b_covs = (double*)(base + o_bv);
  a.b_ams = (double*)(base + o_ba);
  a.b_axes = (double*)(base + o_bx);
  a.b_axl = (double*)(base + o_bl);
  a.b_lv = (double*)(base + o_bg);
  a.records = (double*)(base + o_rec);
  a.fin_ws = (double*)(base + o_fw);
  a.fin_stride = ns_fin_stride(N);
  a.b_cum = (double*)(base + o_cum);
  a.boot_ent = (uint64_t*)(base + o_be);
  a.run_shift = (double*)(base + o_rs);
  a.force_first = (int*)(base + o_ff);
  a.presort = (const unsigned short*)(base + o_pso);
  a.presort_stride = 2048;
  a.presorted = 0;
  a.sel_ent = a.forced_exact ? (uint64_t*)(base + o_se) : nullptr;
  a.undo_u = a.forced_exact ? (double*)(base + o_uu) : nullptr;
  a.undo_slot = a.forced_exact ? (int*)(base + o_us) : nullptr;
  a.fx_kind = a.forced_exact ? (int*)(base + o_fk) : nullptr;
  a.pass_mode = a.forced_exact ? (int*)(base + o_pmk) : nullptr;
  a.fx_pass = 0;
  if (want_pt) {
    a.live_it = (int*)(base + o_lit);
    a.dead_id = (int*)(base + o_pid);
    a.dead_it = (int*)(base + o_pit);
    a.dead_nc = (int*)(base + o_pnc);
  }
  uint32_t* d_ent = (uint32_t*)(base + o_ent);
  hipStream_t s = ctx->stream;
  const int8_t* d_bc = nullptr;  // periodic / reflective coordinates (dh_ns_set_boundary)
  if (!shot.bc.empty()) {
    if (!hip_ok(ctx, hipMemcpyAsync(base + o_bcf, shot.bc.data(), (size_t)D, hipMemcpyHostToDevice, s), "H2D bc"))
      return cleanup(DH_ERR_HIP);
    d_bc = (const int8_t*)(base + o_bcf);
  }
  if (want_pt && !hip_ok(ctx, hipMemsetAsync(base + o_lit, 0, (size_t)R * N * 4, s), "memset")) return cleanup(DH_ERR_HIP);
  if (!hip_ok(ctx, hipMemsetAsync(base + o_nd, 0, 64, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(base + o_bs, 0, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(base + o_fo, 0, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(base + o_ff, 0x7f, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(base + o_us, 0xff, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemsetAsync(base + o_fk, 0, (size_t)R * 4, s), "memset") ||   // fx_kind
      !hip_ok(ctx, hipMemsetAsync(base + o_pmk, 0, (size_t)R * 4, s), "memset") ||  // pass_mode (its own padded block)
      !hip_ok(ctx, hipMemsetAsync(base + o_ne, 0, (size_t)R * 4, s), "memset") ||
      !hip_ok(ctx, hipMemcpyAsync(d_ent, entropy_words, (size_t)n_words * 4, hipMemcpyHostToDevice, s), "H2D"))
    return cleanup(DH_ERR_HIP);

  hipLaunchKernelGGL(ns_init, dim3(R), dim3(kT), 0, s, a, d_ent, n_words, first_run);
  int rc = eval_launch_dev(ctx, problem, R * N, a.live_u, a.live_v, a.live_logl);
  if (rc) return cleanup(rc);
  size_t lds_fin = 64;  // ns_finish: the keys by slot, the keys in order, the sorted slots (large sets: global memory)
  if (!ns_finish_big(N)) {
    size_t Pf = 1;
    while (Pf < (size_t)N) Pf <<= 1;
    lds_fin += (size_t)N * 16 + (Pf > 256 ? Pf : 256) * 2;
  }
  const size_t lds_cons = ns_consume_lds(N, K);
  if (K > kEPTMax * kT) return cleanup(fail(ctx, DH_ERR_ARG, "ns_ensemble: queue_size %d > %d", K, kEPTMax * kT));
  if (lds_cons > 150 * 1024) return cleanup(fail(ctx, DH_ERR_ARG, "ns_ensemble: nlive/queue too large for LDS"));
  const size_t lds_max = lds_cons > lds_fin ? lds_cons : lds_fin;
  if (lds_max > 150 * 1024) return cleanup(fail(ctx, DH_ERR_ARG, "ns_ensemble: nlive/queue too large for LDS"));
  if (me > kMaxCum) return cleanup(fail(ctx, DH_ERR_ARG, "ns_ensemble: nlive/(2 ndim) = %d ellipsoids > %d", me, kMaxCum));
  // the attribute is per device (and the call is cheap): set it on every call, on this context's device
  if (!hip_ok(ctx, hipFuncSetAttribute((const void*)ns_consume_for(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), "hipFuncSetAttribute(ns_consume)") ||
      !hip_ok(ctx, hipFuncSetAttribute((const void*)ns_start, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), "hipFuncSetAttribute(ns_start)") ||
      !hip_ok(ctx, hipFuncSetAttribute((const void*)ns_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max), "hipFuncSetAttribute(ns_finish)"))
    return cleanup(DH_ERR_HIP);
  hipLaunchKernelGGL(ns_start, dim3(R), dim3(kT), 0, s, a);
  if (!hip_ok(ctx, hipGetLastError(), "ns_start launch")) return cleanup(DH_ERR_HIP);
  const int64_t fills_cap = max_fills > 0 ? max_fills : 1000000;
  int64_t fill = 0;
  int ndone = 0;
  int h_state[2] = {0, 1};  // [runs done, any run still in the unit-cube phase]
  bool cube_phase = true;
  const bool force_check = !(getenv("DH_NS_FORCE") && atoi(getenv("DH_NS_FORCE")) == 0);  // diagnostic: 0 = no forced rebuilds
  // bound.update of the runs in rebuild_mask (+ bootstrap expansion, + enlarge): sampler.py:492-508
  auto build_bounds = [&]() -> int {
    int rc = rebuild_launch_masked(ctx, R, a.live_u, N, D, bound_multi ? 0 : 1, me, a.nells, a.bstatus, a.b_ctrs,
                                   a.b_covs, a.b_ams, a.b_axes, a.b_axl, a.b_lv, a.rebuild_mask);
    if (rc) return rc;
    if (bootstrap > 0) {
      // bound.update(points, bootstrap=B): the expansion factor from B resampled replicas per rebuilding run,
      // then scale_to_logvol(logvol + ndim ln(expand)) where it exceeds 1 (bounding.py:381-400, 688-703)
      rc = bootstrap_expand_launch(ctx, R, a.live_u, N, D, bound_multi, me, bootstrap, a.boot_ent, a.rebuild_mask,
                                   base + o_boot, a.run_shift, nullptr, a.bstatus);
      if (rc) return rc;
      rc = enlarge_launch_masked(ctx, R, me, a.nells, D, a.b_covs, a.b_ams, a.b_axes, a.b_axl, a.b_lv, 0.0,
                                 a.rebuild_mask, a.run_shift);
      if (rc) return rc;
    }
    if (enlarge != 1.0)  // sampler.py:506-508
      rc = enlarge_launch_masked(ctx, R, me, a.nells, D, a.b_covs, a.b_ams, a.b_axes, a.b_axl, a.b_lv,
                                 a.enlarge_log, a.rebuild_mask);
    return rc;
  };
  const int n_frames = R * me * (a.forced_exact ? 2 : 1);
  const bool presort_ok = sampler == 0 && N <= 2048 && !ns_consume_compact(N, K) && !a.overlap &&
                          !(getenv("DH_NS_PRESORT") && atoi(getenv("DH_NS_PRESORT")) == 0);
  while (fill < fills_cap && ndone < R) {
    for (int burst = 0; burst < 8 && fill < fills_cap; ++burst, ++fill) {
      if (a.overlap && fill > 0 && !hip_ok(ctx, hipStreamWaitEvent(s, ev_rb, 0), "hipStreamWaitEvent(rebuild)"))
        return cleanup(DH_ERR_HIP);
      a.rebuild_fill = fill % every == 0 ? 1 : 0;
      hipLaunchKernelGGL(ns_prepare, dim3(1), dim3(kT), 0, s, a);
      if (a.overlap) {
        // the bound work of this fill goes to the second stream (everything below enqueues on ctx->stream)
        if (!hip_ok(ctx, hipEventRecord(ev_prep, s), "hipEventRecord") ||
            !hip_ok(ctx, hipStreamWaitEvent(rb_stream, ev_prep, 0), "hipStreamWaitEvent(prepare)"))
          return cleanup(DH_ERR_HIP);
        ctx->stream = rb_stream;
      }
      if (!a.forced_exact) {
        if (a.rebuild_fill) {
          rc = build_bounds();
          if (rc) return cleanup(rc);
        }
        if (a.overlap) {
          ctx->stream = main_stream;
          if (!hip_ok(ctx, hipEventRecord(ev_rb, rb_stream), "hipEventRecord")) return cleanup(DH_ERR_HIP);
        }
        hipLaunchKernelGGL(ns_select, dim3(R), dim3(kT), 0, s, a);
        if (sampler != 3)
          hipLaunchKernelGGL(ns_gather, dim3((unsigned)(((size_t)R * K * D + 255) / 256)), dim3(256), 0, s, a);
        // Sampler.propose_live rebuilds the bound at once when a start point lies outside it (sampler.py:484-489:
        // a point accepted since the last update, beyond the enlarged ellipsoids).  In this form (the opt-in fast one)
        // the run is flagged and rebuilds before its NEXT fill: the walkers of this fill are already chosen, and a
        // queue of K proposals is as stale in the reference.  (Above the register-resident dimensions: one wavefront
        // per start point.)
        if (force_check && sampler != 3) {
          rc = contains_runs_launch(ctx, a.q_u0, R * K, D, K, a.b_ctrs, a.b_ams, bound_multi ? a.nells : nullptr, me,
                                    bound_multi ? 1 : 0, a.run_mode, MODE_BOUND, a.bstatus, a.force, nullptr);
          if (rc) return cleanup(rc);
        }
      } else {
        // The reference's protocol (DH_NS_OPT_FORCED_EXACT, the default of the Python layer since round 5), with ONE
        // rebuild chain per fill that builds bounds.  Pass 1: every run but those of this fill's regular updates
        // selects its queue (a pending run keeps the one it has) and the membership test finds the first start point
        // outside (force_first).  ns_force_prepare: a flagged run takes its forced update with this fill's bounds, or
        // -- in a fill that builds none -- keeps its queue and waits for one that does (its own sequence is unchanged).
        // Then the masked rebuild of the regular AND the forced updates together (the regular ones see the live set
        // without the newest point: ns_swap_undo; the forced ones as it is, their old frames kept: ns_shadow_axes),
        // pass 2 (the regular updates' runs select from their new bounds; the membership test is the reference's check
        // that a forced update worked, and flags a regular update's run whose newest point lies outside its new
        // bound: pending), and ns_reselect (entries behind the first one outside redraw their frames from the new
        // volumes with the same variates).
        const dim3 ggrid((unsigned)(((size_t)R * K * D + 255) / 256));
        const bool starts = sampler != 3;  // (the uniform sampler starts nowhere: no membership test, no forced update)
        a.fx_pass = 1;
        hipLaunchKernelGGL(ns_select, dim3(R), dim3(kT), 0, s, a);
        if (starts) hipLaunchKernelGGL(ns_gather, ggrid, dim3(256), 0, s, a);
        if (force_check && starts) {
          hipLaunchKernelGGL(ns_pass_mask, dim3(1), dim3(kT), 0, s, a, 1);
          rc = contains_runs_launch(ctx, a.q_u0, R * K, D, K, a.b_ctrs, a.b_ams, bound_multi ? a.nells : nullptr, me,
                                    bound_multi ? 1 : 0, a.pass_mode, MODE_BOUND, a.bstatus, a.force, a.force_first);
          if (rc) return cleanup(rc);
          hipLaunchKernelGGL(ns_force_prepare, dim3(1), dim3(kT), 0, s, a);
        }
        if (a.rebuild_fill) {
          if (starts) hipLaunchKernelGGL(ns_shadow_axes, dim3(R, 16), dim3(256), 0, s, a);
          hipLaunchKernelGGL(ns_swap_undo, dim3(R), dim3(64), 0, s, a);
          rc = build_bounds();
          if (rc) return cleanup(rc);
          hipLaunchKernelGGL(ns_swap_undo, dim3(R), dim3(64), 0, s, a);
          a.fx_pass = 2;
          hipLaunchKernelGGL(ns_select, dim3(R), dim3(kT), 0, s, a);
          if (starts) hipLaunchKernelGGL(ns_gather, ggrid, dim3(256), 0, s, a);
          if (force_check && starts) {
            hipLaunchKernelGGL(ns_pass_mask, dim3(1), dim3(kT), 0, s, a, 2);
            rc = contains_runs_launch(ctx, a.q_u0, R * K, D, K, a.b_ctrs, a.b_ams, bound_multi ? a.nells : nullptr, me,
                                      bound_multi ? 1 : 0, a.pass_mode, MODE_BOUND, a.bstatus, a.force, nullptr);
            if (rc) return cleanup(rc);
          }
          if (starts) hipLaunchKernelGGL(ns_reselect, dim3(R), dim3(kT), 0, s, a);
        }
        a.fx_pass = 0;
      }
      // Philox keys: seed from the entropy words (one per stage, so that the stages' offset schemes cannot
      // meet), subsequence = global walker slot (first_run + run) * K + w (independent of the sharding)
      dh::PhiloxKey key;
      key.seed = ((unsigned long long)entropy_words[0] << 32) ^ (n_words > 1 ? entropy_words[1] : 0u) ^ 0x9E3779B97F4A7C15ull;
      key.seq0 = (unsigned long long)first_run * (unsigned long long)K;
      // stages whose consumption depends on the data (unit cube, slice samplers): 2^24 draws per walker and fill
      dh::PhiloxKey key_cube = key, key_slice = key;
      key_cube.seed ^= 0x5BD1E995C0BEull;
      key_slice.seed ^= 0x27D4EB2F511CEull;
      key_cube.offset = key_slice.offset = (unsigned long long)fill << 24;
      if (cube_phase) {
        rc = unif_launch_runs(ctx, problem, R * K, D, D, 0, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr,
                              a.q_rng, 0, a.r_u, a.r_v, a.r_logl, a.r_a, a.r_b, a.q_rng_out, a.run_loglstar,
                              a.run_mode, K, MODE_CUBE, philox ? &key_cube : nullptr);
        if (rc) return cleanup(rc);
      }
      if (sampler == 3) {
        // UniformBoundSampler.sample (internal_samplers.py:243-340): draws from the run's bound until one beats
        // the run's threshold; r_a = calls, r_b = flags
        dh::PhiloxKey key_unif = key_slice;
        key_unif.seed ^= 0x3C6EF372FE94F82Bull;
        rc = unif_launch_runs(ctx, problem, R * K, D, D, R * me, a.b_ctrs, a.b_axes, a.b_ams, a.b_cum, 0.0, d_bc,
                              a.q_rng, 0, a.r_u, a.r_v, a.r_logl, a.r_a, a.r_b, a.q_rng_out, a.run_loglstar,
                              a.run_mode, K, MODE_BOUND, philox ? &key_unif : nullptr, bound_multi ? a.nells : nullptr,
                              me);
      } else if (sampler == 0) {
        // 32-bit draws one walker consumes per fill: per step hiprand_normal4 x ceil(D / 4) and one
        // hiprand_uniform_double (2 draws; padded to 4 so that a fill's block stays 4-aligned)
        key.offset = (unsigned long long)fill * (unsigned long long)walks * (unsigned long long)(4 * ((D + 3) / 4) + 4);
        // the slot order ns_consume needs, sorted in front of the generator pass's grid (beside the walk) where the
        // whole live set is sorted (not the compact form) and fits the generator's LDS
        if (presort_ok) {
          ctx->presort.keys = a.live_logl;
          ctx->presort.out = (unsigned short*)(base + o_pso);
          ctx->presort.n = N;
          ctx->presort.runs = R;
          ctx->presort.stride = a.presort_stride;
          ctx->presort.done = 0;
        }
        rc = rwalk_launch_runs(ctx, problem, R * K, D, D, a.q_u0, a.b_axes, n_frames, a.q_frame, 1.0, 0.0, walks,
                               d_bc, a.q_rng, a.r_u, a.r_v, a.r_logl, a.r_a, a.r_b, a.q_rng_out,
                               a.run_loglstar, a.run_scale, a.run_mode, K, MODE_BOUND, philox ? &key : nullptr);
        a.presorted = presort_ok && ctx->presort.done ? 1 : 0;
        ctx->presort = dh_ctx::PresortReq();
      }
      else
        rc = slice_launch_runs(ctx, problem, R * K, D, sampler - 1, a.q_u0, a.b_axes, n_frames, a.q_frame, 1.0,
                               0.0, walks, 0, a.q_rng, a.r_u, a.r_v, a.r_logl, a.r_a, a.r_b, a.r_c, a.r_d,
                               a.q_rng_out, a.run_loglstar, a.run_scale, a.run_mode, a.run_doubling, K,
                               MODE_BOUND, philox ? &key_slice : nullptr);
      if (rc) return cleanup(rc);
      hipLaunchKernelGGL(ns_consume_for(K), dim3(R), dim3(kT), lds_cons, s, a);
      a.presorted = 0;
    }
    if (!hip_ok(ctx, hipMemcpyAsync(h_state, a.ndone, 8, hipMemcpyDeviceToHost, s), "D2H ndone") ||
        !hip_ok(ctx, hipStreamSynchronize(s), "sync"))
      return cleanup(DH_ERR_HIP);
    ndone = h_state[0];
    if (!h_state[1]) cube_phase = false;  // (as of the last ns_prepare: every run has its first bound)
  }
  if (a.overlap && fill > 0 && !hip_ok(ctx, hipStreamWaitEvent(s, ev_rb, 0), "hipStreamWaitEvent(rebuild)"))
    return cleanup(DH_ERR_HIP);
  hipLaunchKernelGGL(ns_finish, dim3(R), dim3(kT), lds_fin, s, a);
  if (!hip_ok(ctx, hipGetLastError(), "ns launch") ||
      !hip_ok(ctx, hipMemcpyAsync(records, a.records, (size_t)R * 64, hipMemcpyDeviceToHost, s), "D2H records"))
    return cleanup(DH_ERR_HIP);
  if (live_logl_out &&
      !hip_ok(ctx, hipMemcpyAsync(live_logl_out, a.live_logl, (size_t)R * N * 8, hipMemcpyDeviceToHost, s),
              "D2H live"))
    return cleanup(DH_ERR_HIP);
  if (live_u_out &&
      !hip_ok(ctx, hipMemcpyAsync(live_u_out, a.live_u, (size_t)R * N * D * 8, hipMemcpyDeviceToHost, s),
              "D2H live u"))
    return cleanup(DH_ERR_HIP);
  if (want_pt &&
      !hip_ok(ctx, hipMemcpyAsync(live_it_out, a.live_it, (size_t)R * N * 4, hipMemcpyDeviceToHost, s), "D2H live it"))
    return cleanup(DH_ERR_HIP);
  if (dead_u_out || want_pt || dead_logl_out) {
    // only the niter rows each run produced (the caller's runs x max_iter (x ndim) buffers may be
    // far larger than what is touched here: 64 runs x 400 000 would be 300 MB of pageable copies)
    if (!hip_ok(ctx, hipStreamSynchronize(s), "sync")) return cleanup(DH_ERR_HIP);
    for (int r = 0; r < R; ++r) {
      long long nit = (long long)records[(size_t)r * 8 + 2];
      if (nit > a.cap) nit = a.cap;
      if (nit <= 0) continue;
      const size_t o = (size_t)r * a.cap;
      if (dead_logl_out && !hip_ok(ctx, hipMemcpyAsync(dead_logl_out + o, a.dead_logl + o, (size_t)nit * 8,
                                                       hipMemcpyDeviceToHost, s), "D2H dead"))
        return cleanup(DH_ERR_HIP);
      if (dead_u_out && !hip_ok(ctx, hipMemcpyAsync(dead_u_out + o * D, a.dead_u + o * D, (size_t)nit * D * 8,
                                                    hipMemcpyDeviceToHost, s), "D2H dead u"))
        return cleanup(DH_ERR_HIP);
      if (want_pt &&
          (!hip_ok(ctx, hipMemcpyAsync(dead_id_out + o, a.dead_id + o, (size_t)nit * 4, hipMemcpyDeviceToHost, s), "D2H id") ||
           !hip_ok(ctx, hipMemcpyAsync(dead_it_out + o, a.dead_it + o, (size_t)nit * 4, hipMemcpyDeviceToHost, s), "D2H it") ||
           !hip_ok(ctx, hipMemcpyAsync(dead_nc_out + o, a.dead_nc + o, (size_t)nit * 4, hipMemcpyDeviceToHost, s), "D2H nc")))
        return cleanup(DH_ERR_HIP);
    }
  }
  if (n_fills_out) *n_fills_out = fill;
  if (a.prof) {
    long long h[32];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, a.prof, sizeof h, hipMemcpyDeviceToHost);
    fprintf(stderr, "ns_consume cycles (run 0, %lld fills): load+sort %lld | walk %lld | scan %lld | replay+state %lld | dead %lld | live store %lld ; fills integrated serially (all runs): %lld new plateau + %lld carried ; queues of run 0 left to the serial walk: %lld ; parallel walk: ranks %lld | counts %lld | scan+pick %lld | low ranks %lld | merge %lld | slots %lld\n",
            (long long)fill, h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], h[7]);
    fprintf(stderr, "  load+sort = live keys %lld | sort %lld | queue + sums %lld ; scan = ties %lld | weights + ln Z scan %lld | stop %lld | information %lld | rest %lld\n", h[16], h[17], h[0], h[18], h[19], h[20], h[21], h[2]);
    (void)hipFree(a.prof);
  }
  return cleanup(DH_OK);
}

}  // extern "C"
