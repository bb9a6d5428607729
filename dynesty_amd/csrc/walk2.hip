// Proposal kernels, part 2: slice samplers (rslice / slice), uniform sampling
// inside the bound (unif), unit-cube sampling, and the sequential draw used by
// Bound.sample(s).  Same mapping as walk.hip: one walker per lane, state in
// registers, wave-uniform matrices through the scalar cache.
#include <stdlib.h>

#include "ctx.h"
#include "rng_gen.h"

using namespace dh;

namespace {

// ---------------------------------------------------------------------------
// Slice sampling (internal_samplers.py:745-855 RSliceSampler.sample, :593-709
// SliceSampler.sample, :1075-1206 generic_slice_step, :1038-1072 doubling
// acceptance).  The walkers of a wave are at different points of Neal's
// procedure, but every transition needs exactly one evaluation of
//    F(x) = logl(prior_transform(u + x * dir))   (-inf outside the unit cube)
// so the step is written as a per-lane state machine around ONE shared F site.
// ---------------------------------------------------------------------------
enum : int {
  PH_LEFT0 = 0,   // evaluate F(left)  (initial)
  PH_RIGHT0 = 1,  // evaluate F(right) (initial)
  PH_OUT_L = 2,   // stepping out, left edge
  PH_OUT_R = 3,   // stepping out, right edge
  PH_DBL = 4,     // doubling expansion
  PH_SHRINK = 5,  // propose inside [left, right]
  PH_ACC = 6,     // doubling acceptance test (Neal 2003, alg. 6)
  PH_DONE = 7
};

struct SliceArgs {
  ProblemDev prob;
  int k, ndim, slices, m, mode;  // mode 0: rslice, 1: slice (principal axes)
  int doubling0;                 // kwargs['slice_doubling']
  double scale, loglstar;
  const double* u0;
  const double* axes_t;
  const int32_t* axes_idx;
  const uint64_t* rng_in;
  double* u;
  double* v;
  double* logl;
  int32_t* ncalls;
  int32_t* nexpand;
  int32_t* ncontract;
  int32_t* flags;  // bit0: expansion_warning_set, bit1: x == 0 failure
  uint64_t* rng_out;
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  // ensemble form (ns.hip), as in RwalkArgs; run_doubling = per-run slice_doubling
  const double* run_loglstar;
  const double* run_scale;
  const int* run_mode;
  const int* run_doubling;
  int wpr, my_mode;
  PhiloxKey ph;  // RNG_PHILOX
};

template <int N, int KIND, int RNG>
__global__ void __launch_bounds__(64, 2) slice_kernel(SliceArgs a) {
  __shared__ ZigLds zig;
  __shared__ double sx[N * 64];
  __shared__ int sperm[N * 64];
  if constexpr (RNG == RNG_PCG64) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int lane = threadIdx.x;
  const int w = blockIdx.x * 64 + lane;
  const bool live = w < a.k;
  const int wi = live ? w : a.k - 1;
  constexpr int n = N;  // slice samplers require ncdim == ndim (dynesty.py:507-509)

  double u[N], dir[N], acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) u[i] = a.u0[(size_t)wi * n + i];
  LaneGen<RNG> g;
  g.init(a.rng_in, (size_t)wi, &zig, a.ph);
  const int my_frame = a.axes_idx ? a.axes_idx[wi] : 0;

  double loglstar = a.loglstar, scale = a.scale;
  bool doubling = a.doubling0 != 0;
  bool warn_set = false, failed = false;
  bool idle = false;
  if (a.run_mode) {
    const int run = wi / a.wpr;
    idle = a.run_mode[run] != a.my_mode;
    loglstar = a.run_loglstar[run];
    scale = a.run_scale[run];
    doubling = a.run_doubling[run] != 0;
  }
  failed = idle;  // idle lanes never enter the state machine
  int nc = 0, n_expand = 0, n_contract = 0;
  double logl_cur = 0.0;
  const double maxlen = sqrt((double)n) / 2.0;
  const int nsub = a.mode == 0 ? 1 : n;  // slice: one step per principal axis

#pragma unroll 1
  for (int s = 0; s < a.slices; ++s) {
    if (a.mode == 1) {
      // rstate.shuffle(arange(n)): Fisher-Yates from the top (numpy _shuffle_raw)
#pragma unroll 1
      for (int i = 0; i < n; ++i) sperm[i * 64 + lane] = i;
#pragma unroll 1
      for (int i = n - 1; i >= 1; --i) {
        const int j = (int)g.interval((uint64_t)i);
        const int tmp = sperm[i * 64 + lane];
        sperm[i * 64 + lane] = sperm[j * 64 + lane];
        sperm[j * 64 + lane] = tmp;
      }
    }
#pragma unroll 1
    for (int sub = 0; sub < nsub; ++sub) {
      // ---- direction ----
      if (a.mode == 0) {
        double ss = 0.0;
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
          const double x = g.normal();
          sx[i * 64 + lane] = x;
          ss = fma(x, x, ss);
        }
        const double inv = 1.0 / sqrt(ss);  // drhat /= norm(drhat)
#pragma unroll 1
        for (int i = 0; i < n; ++i) sx[i * 64 + lane] = sx[i * 64 + lane] * inv;
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.0;
        bool done = false;
        while (!done) {
          const int cur = __builtin_amdgcn_readfirstlane(my_frame);
          if (cur == my_frame) {
            matvec_sgpr<N>(as_const_uniform(a.axes_t + (size_t)cur * N * N), sx, lane, n, acc);
            done = true;
          }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) dir[i] = acc[i] * scale;  // np.dot(axes, drhat) * scale
      } else {
        // axis = (scale * axes.T)[idx] = scale * axes[:, idx]
        const int idx = sperm[sub * 64 + lane];
        const double* col = a.axes_t + (size_t)my_frame * N * N + (size_t)idx * N;
#pragma unroll
        for (int i = 0; i < N; ++i) dir[i] = scale * col[i];
      }
      // ---- generic_slice_step ----
      const double rand0 = g.uniform();
      double dl = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i) dl = fma(dir[i], dir[i], dl);
      dl = sqrt(dl);
      const double dirnorm = dl > maxlen ? dl / maxlen : 1.0;
#pragma unroll
      for (int i = 0; i < N; ++i) dir[i] = dir[i] / dirnorm;

      double left = -rand0, right = 1.0 - rand0;
      double f_l = 0.0, f_r = 0.0;
      double Lw = 0.0, Rw = 0.0, fLw = 0.0, fRw = 0.0;  // doubling window
      double lhat = 0.0, rhat = 0.0, f_lhat = 0.0, f_rhat = 0.0, x1 = 0.0, logl_x1 = 0.0;
      bool Dflag = false, acc_right = false;
      int Kdbl = 1, nexp_step = 0;
      int phase = failed ? PH_DONE : PH_LEFT0;
      double xq = left;  // abscissa being evaluated

      while (__any(phase != PH_DONE)) {
        const bool act = phase != PH_DONE;
        // F(xq): u_new = u + xq * dir ; unit-cube check ; prior ; likelihood
        double lo = 2.0, hi = -1.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          acc[i] = fma(xq, dir[i], u[i]);
          lo = fmin(lo, acc[i]);
          hi = fmax(hi, acc[i]);
        }
        const bool inside = (lo > 0.0) && (hi < 1.0);
        double f = -INFINITY;
        if (__any(act && inside)) {
          prior_to_lds<N, true, KIND>(a.prob, acc, n, sx, lane);
          const double ll = loglike_lds<N, true, KIND>(a.prob, n, sx, lane, acc);
          if (inside) f = ll;
        }
        if (act) {
          ++nc;
          switch (phase) {
            case PH_LEFT0:
              f_l = f;
              phase = PH_RIGHT0;
              xq = right;
              break;
            case PH_RIGHT0:
              f_r = f;
              if (!doubling) {
                if (f_l > loglstar) {
                  phase = PH_OUT_L;
                  left -= 1.0;
                  xq = left;
                } else if (f_r > loglstar) {
                  phase = PH_OUT_R;
                  right += 1.0;
                  xq = right;
                } else {
                  phase = PH_SHRINK;
                }
              } else {
                phase = PH_DBL;
              }
              break;
            case PH_OUT_L:
              f_l = f;
              ++nexp_step;
              if (f_l > loglstar) {
                left -= 1.0;
                xq = left;
              } else if (f_r > loglstar) {
                phase = PH_OUT_R;
                right += 1.0;
                xq = right;
              } else {
                phase = PH_SHRINK;
              }
              break;
            case PH_OUT_R:
              f_r = f;
              ++nexp_step;
              if (f_r > loglstar) {
                right += 1.0;
                xq = right;
              } else {
                phase = PH_SHRINK;
              }
              break;
            case PH_DBL:
              if (acc_right)
                f_r = f;
              else
                f_l = f;
              nexp_step += Kdbl;
              Kdbl *= 2;
              break;
            case PH_SHRINK: {
              ++n_contract;
              bool ok = f > loglstar;
              if (ok && doubling) {
                // start the acceptance test for x1 = xq
                x1 = xq;
                logl_x1 = f;
                lhat = Lw;
                rhat = Rw;
                f_lhat = fLw;
                f_rhat = fRw;
                Dflag = false;
                phase = PH_ACC;
                ok = false;
              }
              if (ok) {
                logl_cur = f;
#pragma unroll
                for (int i = 0; i < N; ++i) u[i] = fma(xq, dir[i], u[i]);
                phase = PH_DONE;
              } else if (phase == PH_SHRINK) {
                if (xq < 0.0)
                  left = xq;
                else if (xq > 0.0)
                  right = xq;
                else {
                  failed = true;
                  phase = PH_DONE;
                }
              }
              break;
            }
            case PH_ACC:
              if (acc_right)
                f_rhat = f;
              else
                f_lhat = f;
              if (Dflag && loglstar >= f_lhat && loglstar >= f_rhat) {
                // rejected: shrink towards the origin as for any failed proposal
                phase = PH_SHRINK;
                if (x1 < 0.0)
                  left = x1;
                else if (x1 > 0.0)
                  right = x1;
                else {
                  failed = true;
                  phase = PH_DONE;
                }
              }
              break;
            default:
              break;
          }
          // ---- phases that decide their next abscissa after the switch ----
          if (phase == PH_DBL) {
            if (f_l > loglstar || f_r > loglstar) {
              const double V = g.uniform();
              if (V < 0.5) {
                left -= (right - left);
                xq = left;
                acc_right = false;
              } else {
                right += (right - left);
                xq = right;
                acc_right = true;
              }
            } else {
              Lw = left;
              Rw = right;
              fLw = f_l;
              fRw = f_r;
              phase = PH_SHRINK;
            }
          }
          if (phase == PH_ACC) {
            if (rhat - lhat > 1.1) {
              const double M = (lhat + rhat) / 2.0;
              if ((0.0 < M && M <= x1) || (x1 < M && M <= 0.0)) Dflag = true;
              if (x1 < M) {
                rhat = M;
                xq = rhat;
                acc_right = true;
              } else {
                lhat = M;
                xq = lhat;
                acc_right = false;
              }
            } else {
              // accepted
              logl_cur = logl_x1;
#pragma unroll
              for (int i = 0; i < N; ++i) u[i] = fma(x1, dir[i], u[i]);
              phase = PH_DONE;
            }
          }
          if (phase == PH_SHRINK) {
            const double width = right - left;
            xq = left + g.uniform() * width;
          }
        }
      }
      n_expand += nexp_step;
      if (!doubling && nexp_step > 1000) {  // n_expand_threshold (:1096, 1142-1145)
        doubling = true;
        warn_set = true;
      }
    }
  }
  prior_to_lds<N, true, KIND>(a.prob, u, n, sx, lane);
  if (idle) failed = false;
  if (live && !idle) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      a.u[(size_t)w * n + i] = u[i];
      a.v[(size_t)w * n + i] = sx[i * 64 + lane];
    }
    a.logl[w] = logl_cur;
    a.ncalls[w] = nc;
    a.nexpand[w] = n_expand;
    a.ncontract[w] = n_contract;
    a.flags[w] = (warn_set ? 1 : 0) | (failed ? 2 : 0);
    g.store(a.rng_out, (size_t)w);
  }
}

// ---------------------------------------------------------------------------
// UniformBoundSampler.sample (internal_samplers.py:243-340) with the bound's
// sample() inlined: Ellipsoid.sample (bounding.py:307-319) for m == 1,
// MultiEllipsoid.sample (bounding.py:525-590) otherwise; UnitCubeSampler
// (internal_samplers.py:364-441) for m == 0.
// ---------------------------------------------------------------------------
struct UnifArgs {
  ProblemDev prob;
  int k, ndim, ncdim, m;
  double loglstar;
  const double* ctrs;      // m x ncdim
  const double* axes_t;    // m x N x N (padded, transposed)
  const double* ams_p;     // m x N x N (padded precision matrices)
  const double* cumprob;   // m   cumsum(exp(logvol_ells - logvol))
  // RadFriends / SupFriends bound (fr_kind 0 balls, 1 cubes, -1: ellipsoids): m == 1, axes_t =
  // the common shape sqrtm(cov), ams_p = its pseudo-inverse axes_inv (both symmetric, padded)
  int fr_kind, fr_n;
  const uint64_t* rng32_in;  // optional k x 2 {has_uint32, uinteger}: NumPy's buffered half of the 32-bit draws
  uint64_t* rng32_out;       // (integers(n) of the friends bounds consumes it); carried across lock-step rounds
  int propose_only;        // 1: stop at the first candidate inside the cube, no prior / likelihood (lock-step path)
  const double* fr_ctrs;   // fr_n x ncdim centres (the live points)
  const double* fr_ct;     // fr_n x ncdim centres in the whitened frame (ctrs . axes_inv)
  const int8_t* bc;        // ndim or null (nonbounded mask semantics)
  const uint64_t* rng_in;
  int64_t max_tries;     // guard against a bound that cannot reach loglstar
  double* u;
  double* v;
  double* logl;
  int32_t* ncalls;
  int32_t* flags;  // bit0: q == 0 failure (RuntimeError), bit1: max_tries hit
  uint64_t* rng_out;
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
  // ensemble form (ns.hip), as in RwalkArgs
  // (a wavefront serves walkers of ONE run there: run = blockIdx / ceil(wpr / 64))
  const double* run_loglstar;
  const int* run_mode;
  int wpr, my_mode;
  // per-run bounds of the ensemble form: run r owns the ellipsoids [r * run_me, r * run_me + M_r) of ctrs /
  // axes_t / ams_p / cumprob, M_r = run_nells[r] (null: 1); run_me == 0: one bound (or the unit cube) for all
  const int* run_nells = nullptr;
  int run_me = 0;
  PhiloxKey ph;  // RNG_PHILOX
};

template <int N, bool FULL, int KIND, int RNG>
__global__ void __launch_bounds__(64, 2) unif_kernel(UnifArgs a) {
  __shared__ ZigLds zig;
  __shared__ double sx[N * 64];
  const int lane = threadIdx.x;
  int w = blockIdx.x * 64 + lane;
  bool live = w < a.k;
  double loglstar = a.loglstar;
  int M = a.m;      // ellipsoids of this wavefront's bound (0: unit cube)
  size_t eb = 0;    // first of them
  if (a.run_mode) {
    // ensemble form: a wavefront of a run that does not belong to this launch leaves before it stages
    // anything (the unit-cube launch of the resident loop cost 52 us per fill long after the last run had
    // left that phase)
    const int wpw = (a.wpr + 63) >> 6, run = blockIdx.x / wpw, iw = (blockIdx.x - run * wpw) * 64 + lane;
    if (a.run_mode[run] != a.my_mode) return;
    live = iw < a.wpr;
    w = run * a.wpr + (live ? iw : 0);
    loglstar = a.run_loglstar[run];
    if (a.run_me > 0) {
      M = a.run_nells ? a.run_nells[run] : 1;
      eb = (size_t)run * a.run_me;
    }
  }
  if constexpr (RNG == RNG_PCG64) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int wi = live ? w : (a.run_mode ? w : a.k - 1);
  const int n = FULL ? N : a.ndim, nc = FULL ? N : a.ncdim;
  bool done = false;
  const bool idle = false;
  const double* cum = a.cumprob ? a.cumprob + eb : nullptr;
  LaneGen<RNG> g;
  g.init(a.rng_in, (size_t)wi, &zig, a.ph);
  if constexpr (RNG == RNG_PCG64) {
    if (a.rng32_in) {
      g.g.has32 = (uint32_t)a.rng32_in[(size_t)wi * 2];
      g.g.buf32 = (uint32_t)a.rng32_in[(size_t)wi * 2 + 1];
    }
  }
  double x[N], acc[N];
  int ncall = 0, flags = 0;
  double logl_cur = 0.0;
  int64_t tries = 0;
  const double inv_nc = 1.0 / (double)nc;
  while (__any(!done)) {
    bool cand = false;  // x holds a candidate inside the cube
    if (!done) {
      ++tries;
      if (M == 0) {
        // unit cube: rstate.uniform(size=ndim)
#pragma unroll 1
        for (int i = 0; i < n; ++i) sx[i * 64 + lane] = g.uniform();
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = (FULL || i < n) ? sx[i * 64 + lane] : 0.5;
        cand = true;
      } else {
        bool accept = true;
        if (a.fr_kind >= 0) {
          // RadFriends.sample / SupFriends.sample (bounding.py:795-831, 1066-1101): a point of the
          // common shape around a random centre, kept with probability 1/q, q = number of shapes
          // that contain it.  Draw order: balls nc normals + 1 uniform, cubes nc uniforms; then
          // integers(n) (buffered 32-bit Lemire) when n > 1; then 1 uniform iff q > 1.
          double fac = 1.0;
          if (a.fr_kind == 0) {
            double ss = 0.0;
#pragma unroll 1
            for (int i = 0; i < nc; ++i) {
              const double z = g.normal();
              sx[i * 64 + lane] = z;
              ss = fma(z, z, ss);
            }
            fac = pow(g.uniform(), inv_nc) / sqrt(ss);
          } else {
#pragma unroll 1
            for (int i = 0; i < nc; ++i) sx[i * 64 + lane] = -1.0 + 2.0 * g.uniform();
          }
          int idx = 0;
          if (a.fr_n > 1) idx = (int)g.bounded32((uint32_t)(a.fr_n - 1));
#pragma unroll
          for (int i = 0; i < N; ++i) acc[i] = 0.0;
          matvec_sgpr<N>(as_const(a.axes_t), sx, lane, nc, acc);
          const double* c = a.fr_ctrs + (size_t)idx * nc;
#pragma unroll
          for (int i = 0; i < N; ++i) x[i] = (FULL || i < nc) ? fma(fac, acc[i], c[i]) : 0.5;
          if (a.fr_n > 1) {
            // whitened candidate y = x . axes_inv, then brute force over the centres
            // (wave-uniform rows through the scalar cache)
#pragma unroll
            for (int i = 0; i < N; ++i)
              if (FULL || i < nc) sx[i * 64 + lane] = x[i];
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] = 0.0;
            matvec_sgpr<N>(as_const(a.ams_p), sx, lane, nc, acc);
            int q = 0;
            for (int j = 0; j < a.fr_n; ++j) {
              cdptr cj = as_const(a.fr_ct + (size_t)j * nc);
              double sd = 0.0;
#pragma unroll
              for (int i = 0; i < N; ++i) {
                if (FULL || i < nc) {
                  const double e = cj[i] - acc[i];
                  sd = a.fr_kind == 0 ? fma(e, e, sd) : fmax(sd, fabs(e));
                }
              }
              q += (a.fr_kind == 0 ? sqrt(sd) : sd) <= 1.0 ? 1 : 0;
            }
            // q == 0: rounding put the draw outside its own shape (the reference divides by
            // zero there); it is simply redrawn
            if (q == 0)
              accept = false;
            else if (q > 1)
              accept = g.uniform() < (1.0 / (double)q);
          }
        } else {
        int idx = 0;
        if (M > 1) {
          // rand_choice (bounding.py:1300-1308): searchsorted(cumsum(pb), U)
          const double xr = g.uniform();
          while (idx < M - 1 && cum[idx] < xr) ++idx;
        }
        // randsphere: nc normals then one uniform
        double ss = 0.0;
#pragma unroll 1
        for (int i = 0; i < nc; ++i) {
          const double z = g.normal();
          sx[i * 64 + lane] = z;
          ss = fma(z, z, ss);
        }
        const double fac = pow(g.uniform(), inv_nc) / sqrt(ss);
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = 0.0;
        bool mv = false;
        while (!mv) {
          const int cur = __builtin_amdgcn_readfirstlane(idx);
          if (cur == idx) {
            matvec_sgpr<N>(as_const_uniform(a.axes_t + (eb + cur) * N * N), sx, lane, nc, acc);
            mv = true;
          }
        }
        const double* c = a.ctrs + (eb + idx) * nc;
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = (FULL || i < nc) ? fma(fac, acc[i], c[i]) : 0.5;
        if (M > 1) {
          // q = number of ellipsoids containing x (strict), 1/q acceptance
          int q = 0, qloose = 0;
          for (int e = 0; e < M; ++e) {
            cdptr ce = as_const(a.ctrs + (eb + e) * nc);
            cdptr A = as_const(a.ams_p + (eb + e) * N * N);
            double quad = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
              if (FULL || i < nc) {
                double r = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j)
                  if (FULL || j < nc) r = fma(A[i * N + j], x[j] - ce[j], r);
                quad = fma(x[i] - ce[i], r, quad);
              }
            }
            q += quad < 1.0 ? 1 : 0;
            qloose += quad <= 1.0 + 1e-3 ? 1 : 0;
          }
          if (q == 0) {
            q = qloose;
            if (q == 0) {
              flags |= 1;
              done = true;
              accept = false;
            }
          }
          if (accept && q > 1) accept = g.uniform() < (1.0 / (double)q);
        }
        }
        if (accept) {
          // unitcheck(u, nonbounded[:n_cluster])
          bool inside = true;
#pragma unroll
          for (int i = 0; i < N; ++i) {
            if (FULL || i < nc) {
              const int b = a.bc ? a.bc[i] : 0;
              if (b == DH_BC_HARD)
                inside = inside && (x[i] > 0.0) && (x[i] < 1.0);
              else
                inside = inside && (x[i] > -0.5) && (x[i] < 1.5);
            }
          }
          if (inside) {
            if (!FULL) {
              // non-cluster dims: rstate.uniform(size=ndim - n_cluster)
#pragma unroll 1
              for (int i = nc; i < n; ++i) sx[i * 64 + lane] = g.uniform();
#pragma unroll
              for (int i = 0; i < N; ++i)
                if (i >= nc && i < n) x[i] = sx[i * 64 + lane];
            }
            cand = true;
          }
        }
      }
    }
    if (a.propose_only) {
      // lock-step path (arbitrary host likelihood): hand the candidate back
      if (cand) {
        done = true;
        if (live) {
#pragma unroll
          for (int i = 0; i < N; ++i)
            if (FULL || i < n) a.u[(size_t)w * n + i] = x[i];
        }
      }
    } else if (__any(cand)) {
      prior_to_lds<N, FULL, KIND>(a.prob, x, n, sx, lane);
      const double ll = loglike_lds<N, FULL, KIND>(a.prob, n, sx, lane, acc);
      if (cand) {
        ++ncall;
        if (ll > loglstar) {
          logl_cur = ll;
          done = true;
          if (live) {
#pragma unroll
            for (int i = 0; i < N; ++i)
              if (FULL || i < n) {
                a.u[(size_t)w * n + i] = x[i];
                a.v[(size_t)w * n + i] = sx[i * 64 + lane];
              }
          }
        }
      }
    }
    if (!done && tries >= a.max_tries) {
      flags |= 2;
      done = true;
    }
  }
  if (live && !idle) {
    if (!a.propose_only) {
      a.logl[w] = logl_cur;
      a.ncalls[w] = ncall;
    }
    a.flags[w] = flags;
    g.store(a.rng_out, (size_t)w);
    if constexpr (RNG == RNG_PCG64) {
      if (a.rng32_out) {
        a.rng32_out[(size_t)w * 2] = g.g.has32;
        a.rng32_out[(size_t)w * 2 + 1] = g.g.buf32;
      }
    }
  }
}

// ---- UnitCubeSampler.sample with FOUR lanes per walker (internal_samplers.py:364-441) ---------------------------
// The unit-cube phase of the resident loop draws until a point beats the run's threshold: a chain of independent
// tries whose length is geometric, so the launch lasts as long as its unluckiest walker (a hundred tries at the end of
// the phase, 0.38 ms per fill of 64 x 512 walkers with one walker per lane, on a half-empty chip).  Try k of a
// walker consumes draws [k n, (k + 1) n) of the walker's PCG64 stream whatever happened before, so four lanes take
// tries 4 r + t of round r from generators jumped t n draws ahead (the 128-bit LCG jumps: state after m steps =
// A_m s + G_m inc), the first lane that succeeds is the walker's result, its generator state the walker's new
// state and 4 r + t + 1 its call count: the same point, counts and stream as the sequential form, bit for bit
// (tests/test_gpu_slice_unif.py).
struct CubeQArgs {
  ProblemDev prob;
  int k, ndim;
  const uint64_t* rng_in;
  int64_t max_tries;
  double* u;
  double* v;
  double* logl;
  int32_t* ncalls;
  int32_t* flags;
  uint64_t* rng_out;
  const double* run_loglstar;
  const int* run_mode;
  int wpr, my_mode;
  double loglstar;
  PhiloxKey ph;  // RNG_PHILOX
};

// RNG_PHILOX (round 6): the throughput mode had no four-lane form, so the resident loop's unit-cube phase ran one
// walker per lane -- 0.74 ms per fill against 0.12 -- and the "throughput" mode was 5 % slower end to end than the
// parity mode.  Counter based, the jump is free: try k of walker w is the 2 n words from offset + 2 n k of the stream
// keyed (seed, seq0 + w) -- exactly what the lane kernel's sequential tries consume, so both forms give the same
// point, count and (none) state for the same key.
template <int N, bool FULL, int KIND, int RNG = RNG_PCG64>
__global__ void __launch_bounds__(64, 2) cube_quad_kernel(CubeQArgs a) {
  __shared__ double sx[N * 64];
  const int lane = threadIdx.x, t = lane & 3;
  const int w0 = blockIdx.x * 16 + (lane >> 2);
  const bool live = w0 < a.k;
  const int w = live ? w0 : a.k - 1;
  const int n = FULL ? N : a.ndim;
  double loglstar = a.loglstar;
  bool done = false;
  if (a.run_mode) {
    const int run = w / a.wpr;
    if (a.run_mode[run] != a.my_mode) done = true;
    loglstar = a.run_loglstar[run];
  }
  if (!__any(!done)) return;
  const bool idle = done;
  // jump constants: n draws (one try), and from the end of a try to the start of this lane's next one (3 n draws)
  U128 An = {0ull, 1ull}, Gn = {0ull, 0ull};
  Pcg64 g;
  U128 A3 = {0ull, 1ull}, C3 = {0ull, 0ull};
  hiprandStatePhilox4_32_10_t st;
  if constexpr (RNG == RNG_PCG64) {
    {
      const U128 m = {DH_PCG_MULT_HI, DH_PCG_MULT_LO}, one = {0ull, 1ull};
      for (int i = 0; i < n; ++i) {
        An = mul128(An, m);
        Gn = add128(mul128(Gn, m), one);
      }
    }
    g.load(a.rng_in + (size_t)w * 4);
    const U128 Cn = mul128(Gn, g.inc);
    for (int i = 0; i < t; ++i) g.state = add128(mul128(g.state, An), Cn);  // this lane's first try starts t n draws in
    const U128 A2 = mul128(An, An);
    A3 = mul128(A2, An);
    C3 = add128(mul128(add128(mul128(Cn, An), Cn), An), Cn);  // ((Cn An) + Cn) An + Cn
  } else {
    hiprand_init(a.ph.seed, a.ph.seq0 + (unsigned long long)w, a.ph.offset + 2ull * n * t, &st);
  }
  double x[N], acc[N];
  int64_t round = 0;
  int flags = 0;
  while (__any(!done)) {
    bool ok = false;
    double ll = 0.0;
    if (!done) {
#pragma unroll 1
      for (int i = 0; i < n; ++i) {
        if constexpr (RNG == RNG_PCG64)
          sx[i * 64 + lane] = g.next_double();
        else
          sx[i * 64 + lane] = 1.0 - hiprand_uniform_double(&st);  // (LaneGen<RNG_PHILOX>::uniform)
      }
#pragma unroll
      for (int i = 0; i < N; ++i) x[i] = (FULL || i < n) ? sx[i * 64 + lane] : 0.5;
    }
    if (__any(!done)) {
      prior_to_lds<N, FULL, KIND>(a.prob, x, n, sx, lane);
      ll = loglike_lds<N, FULL, KIND>(a.prob, n, sx, lane, acc);
      ok = !done && ll > loglstar;
    }
    // the first of the walker's four lanes that succeeded
    const unsigned long long b = __ballot(ok);
    const unsigned nib = (unsigned)(b >> (lane & ~3)) & 0xfu;
    if (!done && nib) {
      const int first = __ffs((int)nib) - 1;
      if (t == first && live && !idle) {
#pragma unroll
        for (int i = 0; i < N; ++i)
          if (FULL || i < n) {
            a.u[(size_t)w0 * n + i] = x[i];
            a.v[(size_t)w0 * n + i] = sx[i * 64 + lane];
          }
        a.logl[w0] = ll;
        a.ncalls[w0] = (int32_t)(round * 4 + t + 1);
        a.flags[w0] = 0;
        if constexpr (RNG == RNG_PCG64) {
          g.has32 = 0;
          g.buf32 = 0;
          if (a.rng_out) g.store(a.rng_out + (size_t)w0 * 4);
        }
      }
      done = true;
    }
    if (!done) {
      ++round;
      if (round * 4 >= a.max_tries) {
        flags = 2;
        if (t == 0 && live && !idle) {
          a.flags[w0] = flags;
          a.ncalls[w0] = (int32_t)(round * 4);
          a.logl[w0] = 0.0;
        }
        done = true;
      } else if constexpr (RNG == RNG_PCG64) {
        g.state = add128(mul128(g.state, A3), C3);
      } else {
        hiprand_init(a.ph.seed, a.ph.seq0 + (unsigned long long)w, a.ph.offset + 2ull * n * (4ull * (unsigned long long)round + t), &st);
      }
    }
  }
}

// pad + transpose m matrices (row-major nc x nc) to m x N x N; transpose=0 keeps
// the orientation (precision matrices), 1 transposes (frames)
__global__ void pad_mats_kernel(const double* __restrict__ in, int m, int nc, int N, int transpose,
                                double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = m * N * N;
  if (t >= tot) return;
  const int f = t / (N * N), r = t % (N * N), a = r / N, b = r % N;
  const int i = transpose ? b : a, j = transpose ? a : b;
  out[t] = (a < nc && b < nc) ? in[(size_t)f * nc * nc + i * nc + j] : 0.0;
}

// ---------------------------------------------------------------------------
// Bound.sample / samples from ONE generator (Ellipsoid.sample bounding.py:307-
// 334, MultiEllipsoid.sample :525-606): sequential by construction, run by a
// single lane; runtime dimension, operands in global memory.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
    bound_draw_kernel(const uint64_t* state_in, int nsamp, int d, int m, const double* ctrs,
                      const double* axes, const double* ams, const double* cumprob, int return_q,
                      double* xs, int32_t* idxs, int32_t* qs, int32_t* status, uint64_t* state_out,
                      double* work, const uint64_t* zki, const uint64_t* zwi, const uint64_t* zfi) {
  __shared__ ZigLds zig;
  zig_stage(&zig, zki, zwi, zfi);
  if (threadIdx.x != 0) return;
  Pcg64 g;
  g.load(state_in);
  double* z = work;      // d
  double* x = work + d;  // d
  int st = 0;
  for (int s = 0; s < nsamp && st == 0; ++s) {
    for (;;) {
      int idx = 0;
      if (m > 1) {
        const double xr = g.next_double();
        while (idx < m - 1 && cumprob[idx] < xr) ++idx;
      }
      double ss = 0.0;
      for (int i = 0; i < d; ++i) {
        z[i] = std_normal(g, &zig);
        ss = fma(z[i], z[i], ss);
      }
      const double fac = pow(g.next_double(), 1.0 / (double)d) / sqrt(ss);
      const double* A = axes + (size_t)idx * d * d;
      for (int i = 0; i < d; ++i) {
        double r = 0.0;
        for (int j = 0; j < d; ++j) r = fma(A[i * d + j], z[j], r);
        x[i] = fma(fac, r, ctrs[(size_t)idx * d + i]);
      }
      int q = 1;
      if (m > 1) {
        q = 0;
        int qloose = 0;
        for (int e = 0; e < m; ++e) {
          const double* P = ams + (size_t)e * d * d;
          const double* c = ctrs + (size_t)e * d;
          double quad = 0.0;
          for (int i = 0; i < d; ++i) {
            double r = 0.0;
            for (int j = 0; j < d; ++j) r = fma(P[i * d + j], x[j] - c[j], r);
            quad = fma(x[i] - c[i], r, quad);
          }
          q += quad < 1.0 ? 1 : 0;
          qloose += quad <= 1.0 + 1e-3 ? 1 : 0;
        }
        if (q == 0) {
          q = qloose;
          if (q == 0) {
            st = DH_ERR_QZERO;
            break;
          }
        }
      }
      bool take = true;
      if (m > 1 && !return_q) take = (q == 1) || (g.next_double() < (1.0 / (double)q));
      if (take) {
        for (int i = 0; i < d; ++i) xs[(size_t)s * d + i] = x[i];
        idxs[s] = idx;
        qs[s] = q;
        break;
      }
    }
  }
  *status = st;
  g.store(state_out);
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

int ensure_axes_t(dh_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->axes_t_cap) return DH_OK;
  if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
  if (ctx->axes_t) (void)hipFree(ctx->axes_t);
  ctx->axes_t = nullptr;
  ctx->axes_t_cap = 0;
  if (!hip_ok(ctx, hipMalloc((void**)&ctx->axes_t, bytes * 2), "hipMalloc(axes_t)")) return DH_ERR_NOMEM;
  ctx->axes_t_cap = bytes * 2;
  return DH_OK;
}

}  // namespace

extern "C" {

int dh_slice_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                       const double* axes, int m, const int32_t* axes_idx, double scale,
                       double loglstar, int slices, int doubling, const uint64_t* rng, double* u,
                       double* v, double* logl, int32_t* ncalls, int32_t* nexpand, int32_t* ncontract,
                       int32_t* flags, uint64_t* rng_out) {
  return dh::slice_launch_runs(ctx, problem, k, ndim, mode, u0, axes, m, axes_idx, scale, loglstar, slices,
                               doubling, rng, u, v, logl, ncalls, nexpand, ncontract, flags, rng_out, nullptr,
                               nullptr, nullptr, nullptr, 1, 0);
}

int dh_slice_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                              const double* axes, int m, const int32_t* axes_idx, double scale, double loglstar,
                              int slices, int doubling, uint64_t seed, uint64_t sequence0, uint64_t offset,
                              double* u, double* v, double* logl, int32_t* ncalls, int32_t* nexpand,
                              int32_t* ncontract, int32_t* flags) {
  const dh::PhiloxKey key = {seed, sequence0, offset};
  return dh::slice_launch_runs(ctx, problem, k, ndim, mode, u0, axes, m, axes_idx, scale, loglstar, slices,
                               doubling, nullptr, u, v, logl, ncalls, nexpand, ncontract, flags, nullptr, nullptr,
                               nullptr, nullptr, nullptr, 1, 0, &key);
}

}  // extern "C"

int dh::slice_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                          const double* axes, int m, const int32_t* axes_idx, double scale,
                          double loglstar, int slices, int doubling, const uint64_t* rng, double* u,
                          double* v, double* logl, int32_t* ncalls, int32_t* nexpand, int32_t* ncontract,
                          int32_t* flags, uint64_t* rng_out, const double* run_loglstar,
                          const double* run_scale, const int* run_mode, const int* run_doubling, int wpr,
                          int my_mode, const dh::PhiloxKey* philox) {
  DH_CHECK_CTX(ctx);
  SliceArgs a;
  a.ph = philox ? *philox : dh::PhiloxKey{0, 0, 0};
  if (!philox && !rng && k > 0) return fail(ctx, DH_ERR_ARG, "slice: no generator states");
  a.run_loglstar = run_loglstar;
  a.run_scale = run_scale;
  a.run_mode = run_mode;
  a.run_doubling = run_doubling;
  a.wpr = wpr;
  a.my_mode = my_mode;
  if (!get_problem(ctx, problem, &a.prob)) return DH_ERR_ARG;
  if (a.prob.ndim != ndim) return fail(ctx, DH_ERR_ARG, "problem ndim %d != %d", a.prob.ndim, ndim);
  if (k <= 0) return DH_OK;
  if (m < 1 || slices < 1 || (mode != 0 && mode != 1))
    return fail(ctx, DH_ERR_ARG, "slice: m=%d slices=%d mode=%d", m, slices, mode);
  const int N = pad_dim(ndim);
  if (N != ndim)  // no register-resident instantiation for this dimension: wave-per-walker path
    return wide_walk_launch(ctx, mode + 1, problem, k, ndim, ndim, u0, axes, m, axes_idx, scale, loglstar,
                            slices, doubling, nullptr, rng, u, v, logl, ncalls, nexpand, ncontract, flags,
                            rng_out, run_loglstar, run_scale, run_mode, run_doubling, wpr, my_mode, philox);
  int rc = ensure_axes_t(ctx, (size_t)m * N * N * 8);
  if (rc) return rc;
  hipLaunchKernelGGL(pad_mats_kernel, dim3((m * N * N + 255) / 256), dim3(256), 0, ctx->stream, axes, m,
                     ndim, N, 1, ctx->axes_t);
  a.k = k;
  a.ndim = ndim;
  a.slices = slices;
  a.m = m;
  a.mode = mode;
  a.doubling0 = doubling;
  a.scale = scale;
  a.loglstar = loglstar;
  a.u0 = u0;
  a.axes_t = ctx->axes_t;
  a.axes_idx = axes_idx;
  a.rng_in = rng;
  a.u = u;
  a.v = v;
  a.logl = logl;
  a.ncalls = ncalls;
  a.nexpand = nexpand;
  a.ncontract = ncontract;
  a.flags = flags;
  a.rng_out = rng_out;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  const dim3 grid((k + 63) / 64), block(64);
  const int kind = problem_kind(a.prob.like_id, a.prob.prior_id);
  // throughput mode: built for the generic kind of every dimension and for BASELINE C3's (2-D eggbox)
#define LP(NN, KK) hipLaunchKernelGGL((slice_kernel<NN, KK, RNG_PHILOX>), grid, block, 0, ctx->stream, a)
#define X(NN)                                        \
  if (philox && N == NN) {                           \
    if (NN == 2 && kind == KIND_EGGBOX_IDENTITY)     \
      LP(2, KIND_EGGBOX_IDENTITY);                   \
    else                                             \
      LP(NN, KIND_GENERIC);                          \
  }
  DH_DIM_LIST(X)
#undef X
#undef LP
  if (philox) return hip_ok(ctx, hipGetLastError(), "slice launch") ? DH_OK : DH_ERR_HIP;
#define L(NN, KK) hipLaunchKernelGGL((slice_kernel<NN, KK, RNG_PCG64>), grid, block, 0, ctx->stream, a)
#define X(NN)                              \
  if (N == NN) {                           \
    if (kind == KIND_PREC_AFFINE)          \
      L(NN, KIND_PREC_AFFINE);             \
    else if (kind == KIND_IID_AFFINE)      \
      L(NN, KIND_IID_AFFINE);              \
    else if (kind == KIND_EGGBOX_IDENTITY) \
      L(NN, KIND_EGGBOX_IDENTITY);         \
    else if (kind == KIND_IID_NORMAL)      \
      L(NN, KIND_IID_NORMAL);              \
    else                                   \
      L(NN, KIND_GENERIC);                 \
  }
  DH_DIM_LIST(X)
#undef X
#undef L
  return hip_ok(ctx, hipGetLastError(), "slice launch") ? DH_OK : DH_ERR_HIP;
}

extern "C" {


}  // extern "C"

namespace {
// host-pointer form of the batched slice samplers; key == nullptr: PCG64 streams from `rng`
int slice_batch_host(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0, const double* axes, int m,
                     const int32_t* axes_idx, double scale, double loglstar, int slices, int doubling,
                     const uint64_t* rng, double* u, double* v, double* logl, int32_t* ncalls, int32_t* nexpand,
                     int32_t* ncontract, int32_t* flags, uint64_t* rng_out, const dh::PhiloxKey* key) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!u0 || !axes || (!rng && !key) || !u || !v || !logl || !ncalls || !nexpand || !ncontract || !flags)
    return fail(ctx, DH_ERR_ARG, "slice: null pointer");
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  int rc = arena_reserve(ctx, 3 * kd * 8 + (size_t)m * ndim * ndim * 8 + (size_t)k * (8 + 20 + 64) + 8192);
  if (rc) return rc;
  const double* d_u0 = arena_up(ctx, u0, kd);
  const double* d_axes = arena_up(ctx, axes, (size_t)m * ndim * ndim);
  const int32_t* d_idx = axes_idx ? arena_up(ctx, axes_idx, (size_t)k) : nullptr;
  const uint64_t* d_rng = key ? nullptr : arena_up(ctx, rng, (size_t)k * 4);
  double* d_u = (double*)arena_get(ctx, kd * 8);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_l = (double*)arena_get(ctx, (size_t)k * 8);
  int32_t* d_nc = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_ne = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_nt = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_fl = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_ro = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  if (!d_u0 || !d_axes || (!key && !d_rng) || !d_u || !d_v || !d_l || !d_nc || !d_ne || !d_nt || !d_fl || !d_ro ||
      (axes_idx && !d_idx))
    return DH_ERR_NOMEM;
  rc = dh::slice_launch_runs(ctx, problem, k, ndim, mode, d_u0, d_axes, m, d_idx, scale, loglstar, slices, doubling,
                             d_rng, d_u, d_v, d_l, d_nc, d_ne, d_nt, d_fl, key ? nullptr : d_ro, nullptr, nullptr,
                             nullptr, nullptr, 1, 0, key);
  if (rc) return rc;
  if (!down(ctx, u, d_u, kd) || !down(ctx, v, d_v, kd) || !down(ctx, logl, d_l, (size_t)k) ||
      !down(ctx, ncalls, d_nc, (size_t)k) || !down(ctx, nexpand, d_ne, (size_t)k) ||
      !down(ctx, ncontract, d_nt, (size_t)k) || !down(ctx, flags, d_fl, (size_t)k) ||
      (!key && !down(ctx, rng_out, d_ro, (size_t)k * 4)))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  for (int i = 0; i < k; ++i)
    if (flags[i] & 2)
      return fail(ctx, DH_ERR_SLICE, "Slice sampler has failed to find a valid point (walker %d)", i);
  return DH_OK;
}
}  // namespace

extern "C" {

int dh_slice_batch(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0, const double* axes, int m,
                   const int32_t* axes_idx, double scale, double loglstar, int slices, int doubling,
                   const uint64_t* rng, double* u, double* v, double* logl, int32_t* ncalls, int32_t* nexpand,
                   int32_t* ncontract, int32_t* flags, uint64_t* rng_out) {
  return slice_batch_host(ctx, problem, k, ndim, mode, u0, axes, m, axes_idx, scale, loglstar, slices, doubling, rng, u,
                          v, logl, ncalls, nexpand, ncontract, flags, rng_out, nullptr);
}

int dh_slice_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0, const double* axes,
                          int m, const int32_t* axes_idx, double scale, double loglstar, int slices, int doubling,
                          uint64_t seed, uint64_t sequence0, uint64_t offset, double* u, double* v, double* logl,
                          int32_t* ncalls, int32_t* nexpand, int32_t* ncontract, int32_t* flags) {
  const dh::PhiloxKey key = {seed, sequence0, offset};
  return slice_batch_host(ctx, problem, k, ndim, mode, u0, axes, m, axes_idx, scale, loglstar, slices, doubling,
                          nullptr, u, v, logl, ncalls, nexpand, ncontract, flags, nullptr, &key);
}

int dh_unif_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                      const double* axes, const double* ams, const double* cumprob, double loglstar,
                      const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u, double* v,
                      double* logl, int32_t* ncalls, int32_t* flags, uint64_t* rng_out) {
  return dh::unif_launch_runs(ctx, problem, k, ndim, ncdim, m, ctrs, axes, ams, cumprob, loglstar, bc, rng,
                              max_tries, u, v, logl, ncalls, flags, rng_out, nullptr, nullptr, 1, 0);
}

int dh_unif_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                             const double* axes, const double* ams, const double* cumprob, double loglstar,
                             const int8_t* bc, uint64_t seed, uint64_t sequence0, uint64_t offset, int64_t max_tries,
                             double* u, double* v, double* logl, int32_t* ncalls, int32_t* flags) {
  const dh::PhiloxKey key = {seed, sequence0, offset};
  return dh::unif_launch_runs(ctx, problem, k, ndim, ncdim, m, ctrs, axes, ams, cumprob, loglstar, bc, nullptr,
                              max_tries, u, v, logl, ncalls, flags, nullptr, nullptr, nullptr, 1, 0, &key);
}

}  // extern "C"

namespace {

int unif_dispatch(dh_ctx* ctx, const UnifArgs& a, int N, bool philox = false) {
  const int k = a.k, ndim = a.ndim, ncdim = a.ncdim;
  // ensemble form: whole wavefronts per run (a.wpr walkers each)
  const dim3 grid(a.run_mode ? (k / a.wpr) * ((a.wpr + 63) / 64) : (k + 63) / 64), block(64);
  const bool full = (ndim == N && ncdim == N);
  const int kind = (full && !a.propose_only) ? problem_kind(a.prob.like_id, a.prob.prior_id) : KIND_GENERIC;
  // throughput mode: the generic kind of every dimension, and the BASELINE configs' own
  // (C1 3-D iid Normal, C2 25-D correlated Normal, C3 2-D eggbox)
#define LP(NN, FF, KK) hipLaunchKernelGGL((unif_kernel<NN, FF, KK, RNG_PHILOX>), grid, block, 0, ctx->stream, a)
#define X(NN)                                                 \
  if (philox && N == NN) {                                    \
    if (!full)                                                \
      LP(NN, false, KIND_GENERIC);                            \
    else if (NN == 25 && kind == KIND_PREC_AFFINE)            \
      LP(25, true, KIND_PREC_AFFINE);                         \
    else if (NN == 3 && kind == KIND_IID_AFFINE)              \
      LP(3, true, KIND_IID_AFFINE);                           \
    else if (NN == 2 && kind == KIND_EGGBOX_IDENTITY)         \
      LP(2, true, KIND_EGGBOX_IDENTITY);                      \
    else                                                      \
      LP(NN, true, KIND_GENERIC);                             \
  }
  DH_DIM_LIST(X)
#undef X
#undef LP
  if (philox) return hip_ok(ctx, hipGetLastError(), "unif launch") ? DH_OK : DH_ERR_HIP;
#define L(NN, FF, KK) hipLaunchKernelGGL((unif_kernel<NN, FF, KK, RNG_PCG64>), grid, block, 0, ctx->stream, a)
#define X(NN)                              \
  if (N == NN) {                           \
    if (!full)                             \
      L(NN, false, KIND_GENERIC);          \
    else if (kind == KIND_PREC_AFFINE)     \
      L(NN, true, KIND_PREC_AFFINE);       \
    else if (kind == KIND_IID_AFFINE)      \
      L(NN, true, KIND_IID_AFFINE);        \
    else if (kind == KIND_EGGBOX_IDENTITY) \
      L(NN, true, KIND_EGGBOX_IDENTITY);   \
    else if (kind == KIND_IID_NORMAL)      \
      L(NN, true, KIND_IID_NORMAL);        \
    else                                   \
      L(NN, true, KIND_GENERIC);           \
  }
  DH_DIM_LIST(X)
#undef X
#undef L
  return hip_ok(ctx, hipGetLastError(), "unif launch") ? DH_OK : DH_ERR_HIP;
}

}  // namespace

namespace {
int cube_quad_dispatch(dh_ctx* ctx, const CubeQArgs& a, int N, bool philox) {
  const dim3 grid((a.k + 15) / 16), block(64);
  const bool full = a.ndim == N;
  const int kind = full ? problem_kind(a.prob.like_id, a.prob.prior_id) : KIND_GENERIC;
#define LQ(NN, FF, KK)                                                                                         \
  do {                                                                                                         \
    if (philox)                                                                                                \
      hipLaunchKernelGGL((cube_quad_kernel<NN, FF, KK, RNG_PHILOX>), grid, block, 0, ctx->stream, a);          \
    else                                                                                                       \
      hipLaunchKernelGGL((cube_quad_kernel<NN, FF, KK, RNG_PCG64>), grid, block, 0, ctx->stream, a);           \
  } while (0)
#define X(NN)                                         \
  if (N == NN) {                                      \
    if (!full)                                        \
      LQ(NN, false, KIND_GENERIC);                    \
    else if (NN == 25 && kind == KIND_PREC_AFFINE)    \
      LQ(25, true, KIND_PREC_AFFINE);                 \
    else if (NN == 3 && kind == KIND_IID_AFFINE)      \
      LQ(3, true, KIND_IID_AFFINE);                   \
    else if (NN == 2 && kind == KIND_EGGBOX_IDENTITY) \
      LQ(2, true, KIND_EGGBOX_IDENTITY);              \
    else                                              \
      LQ(NN, true, KIND_GENERIC);                     \
  }
  DH_DIM_LIST(X)
#undef X
#undef LQ
  return hip_ok(ctx, hipGetLastError(), "unit-cube launch") ? DH_OK : DH_ERR_HIP;
}
}  // namespace

int dh::unif_launch_runs(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                         const double* axes, const double* ams, const double* cumprob, double loglstar,
                         const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u, double* v,
                         double* logl, int32_t* ncalls, int32_t* flags, uint64_t* rng_out,
                         const double* run_loglstar, const int* run_mode, int wpr, int my_mode,
                         const dh::PhiloxKey* philox, const int* run_nells, int run_me) {
  DH_CHECK_CTX(ctx);
  UnifArgs a;
  a.run_nells = run_nells;
  a.run_me = run_me;
  if (run_mode && (wpr < 1 || k % wpr)) return fail(ctx, DH_ERR_ARG, "unif: k=%d is not runs x %d", k, wpr);
  a.ph = philox ? *philox : dh::PhiloxKey{0, 0, 0};
  if (!philox && !rng && k > 0) return fail(ctx, DH_ERR_ARG, "unif: no generator states");
  a.run_loglstar = run_loglstar;
  a.run_mode = run_mode;
  a.wpr = wpr;
  a.my_mode = my_mode;
  a.propose_only = problem == -1 ? 1 : 0;
  if (a.propose_only) {
    a.prob = ProblemDev();
    a.prob.ndim = ndim;
    a.prob.like_id = 99;  // never evaluated
    a.prob.prior_id = 99;
  } else {
    if (!get_problem(ctx, problem, &a.prob)) return DH_ERR_ARG;
    if (a.prob.ndim != ndim) return fail(ctx, DH_ERR_ARG, "problem ndim %d != %d", a.prob.ndim, ndim);
  }
  if (k <= 0) return DH_OK;
  if (m < 0 || ncdim < 1 || ncdim > ndim) return fail(ctx, DH_ERR_ARG, "unif: m=%d ncdim=%d", m, ncdim);
  if (ndim > kMaxRegDim) {
    if (m != 0 || a.propose_only) {
      // (round 5: the ensemble form -- per-run thresholds and bounds -- above the register-resident dimensions too)
      return wide_unif_launch(ctx, problem, k, ndim, ncdim, m, ctrs, axes, ams, cumprob, loglstar, bc, rng,
                              max_tries, u, v, logl, ncalls, flags, rng_out, philox, run_loglstar, run_mode, wpr,
                              my_mode, run_nells, run_me);
    }
    return wide_walk_launch(ctx, 3, problem, k, ndim, ndim, nullptr, nullptr, 1,
                            nullptr, 1.0, loglstar, 0, 0, bc, rng, u, v, logl, ncalls, nullptr, nullptr,
                            flags, rng_out, run_loglstar, nullptr, run_mode, nullptr, wpr, my_mode, philox);
  }
  const int N = pad_dim(ndim);
  // the unit cube with four lanes per walker (PCG64 streams) where one walker per lane would leave SIMDs empty
  if (m == 0 && !a.propose_only && (philox || rng) &&
      (ctx->cube_form == 2 || (ctx->cube_form == 0 && k <= 256 * ctx->num_cu))) {
    CubeQArgs q;
    q.prob = a.prob;
    q.k = k;
    q.ndim = ndim;
    q.rng_in = rng;
    q.max_tries = max_tries > 0 ? max_tries : ((int64_t)1 << 32);
    q.u = u;
    q.v = v;
    q.logl = logl;
    q.ncalls = ncalls;
    q.flags = flags;
    q.rng_out = rng_out;
    q.run_loglstar = run_loglstar;
    q.run_mode = run_mode;
    q.wpr = wpr;
    q.my_mode = my_mode;
    q.loglstar = loglstar;
    q.ph = a.ph;
    return cube_quad_dispatch(ctx, q, N, philox != nullptr);
  }
  const size_t mats = (size_t)(m > 0 ? m : 1) * N * N * 8;
  int rc = ensure_axes_t(ctx, 2 * mats);
  if (rc) return rc;
  double* at = ctx->axes_t;
  double* ap = ctx->axes_t + (size_t)(m > 0 ? m : 1) * N * N;
  if (m > 0) {
    hipLaunchKernelGGL(pad_mats_kernel, dim3((m * N * N + 255) / 256), dim3(256), 0, ctx->stream, axes, m,
                       ncdim, N, 1, at);
    if (m > 1)
      hipLaunchKernelGGL(pad_mats_kernel, dim3((m * N * N + 255) / 256), dim3(256), 0, ctx->stream, ams,
                         m, ncdim, N, 0, ap);
  }
  a.k = k;
  a.ndim = ndim;
  a.ncdim = ncdim;
  a.m = m;
  a.loglstar = loglstar;
  a.ctrs = ctrs;
  a.axes_t = at;
  a.ams_p = ap;
  a.cumprob = cumprob;
  a.fr_kind = -1;
  a.fr_n = 0;
  a.fr_ctrs = a.fr_ct = nullptr;
  a.rng32_in = nullptr;
  a.rng32_out = nullptr;
  a.bc = bc;
  a.rng_in = rng;
  a.max_tries = max_tries > 0 ? max_tries : ((int64_t)1 << 32);
  a.u = u;
  a.v = v;
  a.logl = logl;
  a.ncalls = ncalls;
  a.flags = flags;
  a.rng_out = rng_out;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  return unif_dispatch(ctx, a, N, philox != nullptr);
}

extern "C" {


}  // extern "C"

namespace {
// host-pointer form of the batched UniformBoundSampler / UnitCubeSampler; key == nullptr: PCG64 streams from `rng`
int unif_batch_host(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs, const double* axes,
                    const double* ams, const double* cumprob, double loglstar, const int8_t* bc, const uint64_t* rng,
                    int64_t max_tries, double* u, double* v, double* logl, int32_t* ncalls, uint64_t* rng_out,
                    const dh::PhiloxKey* key) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if ((!rng && !key) || !u || (problem != -1 && (!v || !logl || !ncalls)) || (m > 0 && (!ctrs || !axes)) ||
      (m > 1 && (!ams || !cumprob)))
    return fail(ctx, DH_ERR_ARG, "unif: null pointer");
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim, mm = (size_t)(m > 0 ? m : 1);
  int rc = arena_reserve(ctx, 2 * kd * 8 + mm * ((size_t)2 * ncdim * ncdim + ncdim + 1) * 8 +
                                  (size_t)k * (8 + 8 + 64) + (size_t)ndim + 8192);
  if (rc) return rc;
  const double* d_c = m > 0 ? arena_up(ctx, ctrs, (size_t)m * ncdim) : nullptr;
  const double* d_ax = m > 0 ? arena_up(ctx, axes, (size_t)m * ncdim * ncdim) : nullptr;
  const double* d_am = m > 1 ? arena_up(ctx, ams, (size_t)m * ncdim * ncdim) : nullptr;
  const double* d_cp = m > 1 ? arena_up(ctx, cumprob, (size_t)m) : nullptr;
  const int8_t* d_bc = bc ? arena_up(ctx, bc, (size_t)ndim) : nullptr;
  const uint64_t* d_rng = key ? nullptr : arena_up(ctx, rng, (size_t)k * 4);
  double* d_u = (double*)arena_get(ctx, kd * 8);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_l = (double*)arena_get(ctx, (size_t)k * 8);
  int32_t* d_nc = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_fl = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_ro = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  if ((!key && !d_rng) || !d_u || !d_v || !d_l || !d_nc || !d_fl || !d_ro) return DH_ERR_NOMEM;
  rc = dh::unif_launch_runs(ctx, problem, k, ndim, ncdim, m, d_c, d_ax, d_am, d_cp, loglstar, d_bc, d_rng, max_tries,
                            d_u, d_v, d_l, d_nc, d_fl, key ? nullptr : d_ro, nullptr, nullptr, 1, 0, key);
  if (rc) return rc;
  std::vector<int32_t> fl((size_t)k);
  if (!down(ctx, u, d_u, kd) || !down(ctx, fl.data(), d_fl, (size_t)k) ||
      (!key && !down(ctx, rng_out, d_ro, (size_t)k * 4)))
    return DH_ERR_HIP;
  if (problem != -1 && (!down(ctx, v, d_v, kd) || !down(ctx, logl, d_l, (size_t)k) ||
                        !down(ctx, ncalls, d_nc, (size_t)k)))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  for (int i = 0; i < k; ++i) {
    if (fl[i] & 1) return fail(ctx, DH_ERR_QZERO, "Ellipsoid check failed q=0 (walker %d)", i);
    if (fl[i] & 2)
      return fail(ctx, DH_ERR_ARG, "unif: walker %d exceeded max_tries without reaching loglstar", i);
  }
  return DH_OK;
}
}  // namespace

extern "C" {

int dh_unif_batch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs, const double* axes,
                  const double* ams, const double* cumprob, double loglstar, const int8_t* bc, const uint64_t* rng,
                  int64_t max_tries, double* u, double* v, double* logl, int32_t* ncalls, uint64_t* rng_out) {
  return unif_batch_host(ctx, problem, k, ndim, ncdim, m, ctrs, axes, ams, cumprob, loglstar, bc, rng, max_tries, u, v,
                         logl, ncalls, rng_out, nullptr);
}

int dh_unif_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                         const double* axes, const double* ams, const double* cumprob, double loglstar, const int8_t* bc,
                         uint64_t seed, uint64_t sequence0, uint64_t offset, int64_t max_tries, double* u, double* v,
                         double* logl, int32_t* ncalls) {
  const dh::PhiloxKey key = {seed, sequence0, offset};
  return unif_batch_host(ctx, problem, k, ndim, ncdim, m, ctrs, axes, ams, cumprob, loglstar, bc, nullptr, max_tries, u,
                         v, logl, ncalls, nullptr, &key);
}

// UniformBoundSampler.sample over a queue with a RadFriends / SupFriends bound
// (internal_samplers.py:243-340 + bounding.py:795-831, 1066-1101); see include/dynhip.h
int dh_unif_friends_batch(dh_ctx* ctx, int problem, int k, int ndim, int kind, const double* ctrs, int n,
                          const double* axes, const double* axes_inv, double loglstar, const int8_t* bc,
                          const uint64_t* rng, int64_t max_tries, double* u, double* v, double* logl,
                          int32_t* ncalls, uint64_t* rng_out, const uint64_t* rng32, uint64_t* rng32_out) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!rng || !u || !ctrs || !axes || !axes_inv || n < 1 || (kind != 0 && kind != 1) ||
      (problem != -1 && (!v || !logl || !ncalls)))
    return fail(ctx, DH_ERR_ARG, "unif_friends: bad arguments");
  UnifArgs a;
  a.propose_only = problem == -1 ? 1 : 0;
  if (a.propose_only) {
    a.prob = ProblemDev();
    a.prob.ndim = ndim;
    a.prob.like_id = 99;
    a.prob.prior_id = 99;
  } else {
    if (!get_problem(ctx, problem, &a.prob)) return DH_ERR_ARG;
    if (a.prob.ndim != ndim) return fail(ctx, DH_ERR_ARG, "problem ndim %d != %d", a.prob.ndim, ndim);
  }
  if (ndim > kMaxRegDim) return fail(ctx, DH_ERR_ARG, "unif_friends: ndim=%d > %d not built", ndim, kMaxRegDim);
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim, nd = (size_t)n * ndim, dd = (size_t)ndim * ndim;
  int rc = arena_reserve(ctx, 2 * kd * 8 + 2 * nd * 8 + 2 * dd * 8 + (size_t)k * (8 + 8 + 64 + 32) + (size_t)ndim + 8192);
  if (rc) return rc;
  const double* d_c = arena_up(ctx, ctrs, nd);
  const double* d_ax = arena_up(ctx, axes, dd);
  const double* d_ai = arena_up(ctx, axes_inv, dd);
  double* d_ct = (double*)arena_get(ctx, nd * 8);
  const int8_t* d_bc = bc ? arena_up(ctx, bc, (size_t)ndim) : nullptr;
  const uint64_t* d_rng = arena_up(ctx, rng, (size_t)k * 4);
  double* d_u = (double*)arena_get(ctx, kd * 8);
  double* d_v = (double*)arena_get(ctx, kd * 8);
  double* d_l = (double*)arena_get(ctx, (size_t)k * 8);
  int32_t* d_nc = (int32_t*)arena_get(ctx, (size_t)k * 4);
  int32_t* d_fl = (int32_t*)arena_get(ctx, (size_t)k * 4);
  uint64_t* d_ro = (uint64_t*)arena_get(ctx, (size_t)k * 32);
  const uint64_t* d_r32 = rng32 ? arena_up(ctx, rng32, (size_t)k * 2) : nullptr;
  uint64_t* d_r32o = rng32_out ? (uint64_t*)arena_get(ctx, (size_t)k * 16) : nullptr;
  if (!d_c || !d_ax || !d_ai || !d_ct || !d_rng || !d_u || !d_v || !d_l || !d_nc || !d_fl || !d_ro ||
      (rng32 && !d_r32) || (rng32_out && !d_r32o))
    return DH_ERR_NOMEM;
  if ((rc = friends_whiten_launch(ctx, d_c, d_ai, n, ndim, d_ct))) return rc;
  const int N = pad_dim(ndim);
  if ((rc = ensure_axes_t(ctx, (size_t)2 * N * N * 8))) return rc;
  double* at = ctx->axes_t;
  double* ap = ctx->axes_t + (size_t)N * N;
  hipLaunchKernelGGL(pad_mats_kernel, dim3((N * N + 255) / 256), dim3(256), 0, ctx->stream, d_ax, 1, ndim, N, 1, at);
  hipLaunchKernelGGL(pad_mats_kernel, dim3((N * N + 255) / 256), dim3(256), 0, ctx->stream, d_ai, 1, ndim, N, 1, ap);
  a.run_loglstar = nullptr;
  a.run_mode = nullptr;
  a.wpr = 1;
  a.my_mode = 0;
  a.k = k;
  a.ndim = ndim;
  a.ncdim = ndim;
  a.m = 1;
  a.loglstar = loglstar;
  a.ctrs = d_c;
  a.axes_t = at;
  a.ams_p = ap;
  a.cumprob = nullptr;
  a.fr_kind = kind;
  a.fr_n = n;
  a.fr_ctrs = d_c;
  a.fr_ct = d_ct;
  a.rng32_in = d_r32;
  a.rng32_out = d_r32o;
  a.bc = d_bc;
  a.rng_in = d_rng;
  a.max_tries = max_tries > 0 ? max_tries : ((int64_t)1 << 32);
  a.u = d_u;
  a.v = d_v;
  a.logl = d_l;
  a.ncalls = d_nc;
  a.flags = d_fl;
  a.rng_out = d_ro;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  if ((rc = unif_dispatch(ctx, a, N))) return rc;
  std::vector<int32_t> fl((size_t)k);
  if (!down(ctx, u, d_u, kd) || !down(ctx, fl.data(), d_fl, (size_t)k) || !down(ctx, rng_out, d_ro, (size_t)k * 4) ||
      (rng32_out && !down(ctx, rng32_out, d_r32o, (size_t)k * 2)))
    return DH_ERR_HIP;
  if (problem != -1 && (!down(ctx, v, d_v, kd) || !down(ctx, logl, d_l, (size_t)k) ||
                        !down(ctx, ncalls, d_nc, (size_t)k)))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  for (int i = 0; i < k; ++i)
    if (fl[i] & 2)
      return fail(ctx, DH_ERR_ARG, "unif_friends: walker %d exceeded max_tries without reaching loglstar", i);
  return DH_OK;
}

int dh_bound_draw(dh_ctx* ctx, const uint64_t* state4, int nsamp, int d, int m, const double* ctrs,
                  const double* axes, const double* ams, const double* cumprob, int return_q, double* xs,
                  int32_t* idxs, int32_t* qs, uint64_t* state4_out) {
  DH_CHECK_CTX(ctx);
  if (nsamp <= 0) return DH_OK;
  if (!state4 || !ctrs || !axes || !xs || !idxs || !qs || !state4_out || m < 1 || d < 1 ||
      (m > 1 && (!ams || !cumprob)))
    return fail(ctx, DH_ERR_ARG, "bound_draw: bad arguments");
  arena_reset(ctx);
  const size_t dd = (size_t)d * d;
  int rc = arena_reserve(ctx, ((size_t)m * (2 * dd + d + 1) + (size_t)nsamp * d + 2 * d) * 8 +
                                  (size_t)nsamp * 8 + 8192);
  if (rc) return rc;
  const uint64_t* d_s = arena_up(ctx, state4, 4);
  const double* d_c = arena_up(ctx, ctrs, (size_t)m * d);
  const double* d_ax = arena_up(ctx, axes, (size_t)m * dd);
  const double* d_am = m > 1 ? arena_up(ctx, ams, (size_t)m * dd) : nullptr;
  const double* d_cp = m > 1 ? arena_up(ctx, cumprob, (size_t)m) : nullptr;
  double* d_x = (double*)arena_get(ctx, (size_t)nsamp * d * 8);
  int32_t* d_i = (int32_t*)arena_get(ctx, (size_t)nsamp * 4);
  int32_t* d_q = (int32_t*)arena_get(ctx, (size_t)nsamp * 4);
  int32_t* d_st = (int32_t*)arena_get(ctx, 4);
  uint64_t* d_o = (uint64_t*)arena_get(ctx, 32);
  double* d_w = (double*)arena_get(ctx, (size_t)2 * d * 8);
  if (!d_s || !d_c || !d_ax || !d_x || !d_i || !d_q || !d_st || !d_o || !d_w) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(bound_draw_kernel, dim3(1), dim3(64), 0, ctx->stream, d_s, nsamp, d, m, d_c, d_ax,
                     d_am, d_cp, return_q, d_x, d_i, d_q, d_st, d_o, d_w, ctx->zki(), ctx->zwi(),
                     ctx->zfi());
  if (!hip_ok(ctx, hipGetLastError(), "bound_draw launch")) return DH_ERR_HIP;
  int32_t st = 0;
  if (!down(ctx, xs, d_x, (size_t)nsamp * d) || !down(ctx, idxs, d_i, (size_t)nsamp) ||
      !down(ctx, qs, d_q, (size_t)nsamp) || !down(ctx, &st, d_st, 1) || !down(ctx, state4_out, d_o, 4))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  if (st == DH_ERR_QZERO) return fail(ctx, DH_ERR_QZERO, "Ellipsoid check failed q=0");
  return DH_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Slice sampling in lock step with a HOST likelihood (dh_slice_feed).
// With an arbitrary Python likelihood every F(x) of generic_slice_step
// (internal_samplers.py:1076-1206) is a host call, so the host drives the stepping-out /
// shrinking state machine and the device supplies what the walker's stream produces: the
// slice direction (rslice: standard_normal(n), normalised, times the frame and the scale,
// :818-823; slice: the shuffled axis order, :667-669) and then the uniforms the step will
// consume (rand0, one per doubling expansion, one per shrink) as a LOOKAHEAD of `nlook`
// values that is not committed -- the next call first advances the stream by the number
// the host actually used.  One wavefront per walker.
namespace {

struct SliceFeedArgs {
  int k, ndim, kind, m, nlook;
  double scale;
  const double* axes;  // m x ndim x ndim (row-major, as passed to the samplers)
  const int32_t* axes_idx;
  uint64_t* st6;  // k x 6: PCG64 state hi, lo, inc hi, lo, has_uint32, uinteger
  const int32_t* consumed;
  double* dirs;   // kind 0: k x ndim
  int32_t* perm;  // kind 1: k x ndim
  double* look;   // k x nlook
  const uint64_t* zki;
  const uint64_t* zwi;
  const uint64_t* zfi;
};

__global__ void __launch_bounds__(64) slice_feed_kernel(SliceFeedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ ZigLds zig;
  double* dr = (double*)smem;        // ndim
  int* sperm = (int*)(dr + a.ndim);  // ndim
  if (a.kind == 0) zig_stage(&zig, a.zki, a.zwi, a.zfi);
  const int w = blockIdx.x, lane = threadIdx.x, D = a.ndim;
  uint64_t* st = a.st6 + (size_t)w * 6;
  Pcg64 g;
  g.load(st);
  g.has32 = (uint32_t)st[4];
  g.buf32 = (uint32_t)st[5];
  const int adv = a.consumed ? a.consumed[w] : 0;
  for (int i = 0; i < adv; ++i) g.step();
  PcgLanes PL = pcg_lanes_init(g, lane);
  if (a.kind == 0) {
    wave_normals(g, PL, &zig, dr, D, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double ss = 0.0;
    for (int i = 0; i < D; ++i) ss = fma(dr[i], dr[i], ss);  // index order, every lane the same
    const double nrm = sqrt(ss);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < D; i += 64) dr[i] = dr[i] / nrm;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const double* A = a.axes + (size_t)(a.axes_idx ? a.axes_idx[w] : 0) * D * D;
    for (int i = lane; i < D; i += 64) {
      double acc = 0.0;
      for (int j = 0; j < D; ++j) acc = fma(A[(size_t)i * D + j], dr[j], acc);
      a.dirs[(size_t)w * D + i] = acc * a.scale;
    }
  } else if (a.kind == 1) {
    for (int i = lane; i < D; i += 64) sperm[i] = i;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = D - 1; i >= 1; --i) {  // Generator.shuffle: Fisher-Yates from the top
      const int j = (int)g.interval((uint64_t)i);
      if (lane == 0) {
        const int tmp = sperm[i];
        sperm[i] = sperm[j];
        sperm[j] = tmp;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < D; i += 64) a.perm[(size_t)w * D + i] = sperm[i];
  }
  if (lane == 0) {
    g.store(st);
    st[4] = g.has32;
    st[5] = g.buf32;
  }
  if (a.nlook > 0) {
    Pcg64 g2 = g;
    wave_doubles(g2, PL, a.look + (size_t)w * a.nlook, a.nlook, lane);
  }
}

}  // namespace

extern "C" {

int dh_slice_feed(dh_ctx* ctx, int k, int ndim, int kind, const double* axes, int m, const int32_t* axes_idx,
                  double scale, uint64_t* state6, const int32_t* consumed, int nlook, double* dirs,
                  int32_t* perm, double* look) {
  DH_CHECK_CTX(ctx);
  if (k <= 0) return DH_OK;
  if (!state6 || kind < 0 || kind > 2 || ndim < 1 || nlook < 0 || (nlook > 0 && !look) ||
      (kind == 0 && (!axes || !dirs || m < 1)) || (kind == 1 && !perm))
    return fail(ctx, DH_ERR_ARG, "slice_feed: bad argument (kind=%d ndim=%d nlook=%d)", kind, ndim, nlook);
  arena_reset(ctx);
  const size_t kd = (size_t)k * ndim;
  int rc = arena_reserve(ctx, (kind == 0 ? (size_t)m * ndim * ndim * 8 + kd * 8 : 0) + kd * 4 +
                                  (size_t)k * (48 + 4 + 4 + (size_t)nlook * 8) + 8192);
  if (rc) return rc;
  SliceFeedArgs a;
  a.k = k;
  a.ndim = ndim;
  a.kind = kind;
  a.m = m;
  a.nlook = nlook;
  a.scale = scale;
  a.axes = kind == 0 ? arena_up(ctx, axes, (size_t)m * ndim * ndim) : nullptr;
  a.axes_idx = (kind == 0 && axes_idx) ? arena_up(ctx, axes_idx, (size_t)k) : nullptr;
  a.st6 = arena_up(ctx, state6, (size_t)k * 6);
  a.consumed = consumed ? arena_up(ctx, consumed, (size_t)k) : nullptr;
  a.dirs = kind == 0 ? (double*)arena_get(ctx, kd * 8) : nullptr;
  a.perm = kind == 1 ? (int32_t*)arena_get(ctx, kd * 4) : nullptr;
  a.look = nlook ? (double*)arena_get(ctx, (size_t)k * nlook * 8) : nullptr;
  a.zki = ctx->zki();
  a.zwi = ctx->zwi();
  a.zfi = ctx->zfi();
  if (!a.st6 || (kind == 0 && (!a.axes || !a.dirs)) || (kind == 1 && !a.perm) || (nlook && !a.look) ||
      (consumed && !a.consumed) || (kind == 0 && axes_idx && !a.axes_idx))
    return DH_ERR_NOMEM;
  hipLaunchKernelGGL(slice_feed_kernel, dim3(k), dim3(64), (size_t)ndim * 12 + 16, ctx->stream, a);
  if (!hip_ok(ctx, hipGetLastError(), "slice feed launch")) return DH_ERR_HIP;
  if (!down(ctx, state6, (const uint64_t*)a.st6, (size_t)k * 6)) return DH_ERR_HIP;
  if (kind == 0 && !down(ctx, dirs, (const double*)a.dirs, kd)) return DH_ERR_HIP;
  if (kind == 1 && !down(ctx, perm, (const int32_t*)a.perm, kd)) return DH_ERR_HIP;
  if (nlook && !down(ctx, look, (const double*)a.look, (size_t)k * nlook)) return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"
