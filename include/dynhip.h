/* dynhip.h -- C ABI of libdynhip.so, the MI355X (gfx950) implementation of
 * dynesty's bounding + proposal hot path.
 *
 * The reference (joshspeagle/dynesty 3.0.0) is pure Python and has no FFI of
 * its own; the entry points below are what a ctypes binding for this path
 * binds to.  Each one names the reference function(s) it replaces
 * (file:line relative to /root/reference/py/dynesty/).  INTEGRATION.md shows
 * the reference-side stubs; dynesty_amd/_lib.py is the binding used here.
 *
 * Conventions
 *   - plain C types only; all matrices are C-contiguous row-major float64.
 *   - every function returns DH_OK (0) or a negative DH_ERR_* code; the
 *     message is available from dh_last_error(ctx).  The Python shim maps the
 *     codes back to the exception types the reference raises.
 *   - host-pointer entry points copy in, launch on the context's stream, copy
 *     out and synchronise before returning; the library never keeps a host
 *     pointer.  `*_dev` entry points take device pointers (from dh_malloc),
 *     only enqueue work on the context's stream and do NOT synchronise.
 *   - a dh_ctx is bound to one device and one stream; calls on one context are
 *     serialised.  No global state: fork/spawn-safe until the first dh_create.
 *   - RNG: one numpy-compatible PCG64 stream per walker, 4 x uint64 words
 *     {state_hi, state_lo, inc_hi, inc_lo} (numpy PCG64().state['state']).
 */
#ifndef DYNHIP_H
#define DYNHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_VERSION 100

#define DH_OK 0
#define DH_ERR_VALUE (-1)   /* ValueError: single point / singular covariance (bounding.py:1405-1407, 220-224) */
#define DH_ERR_CONTAIN (-2) /* RuntimeError: failed to contain all the points (bounding.py:1451-1453) */
#define DH_ERR_REGION (-3)  /* RuntimeError: Rejecting invalid MultiEllipsoid region (bounding.py:683-685) */
#define DH_ERR_SLICE (-4)   /* RuntimeError: slice sampler failed, x == 0 (internal_samplers.py:1188-1199) */
#define DH_ERR_HIP (-5)     /* HIP runtime failure */
#define DH_ERR_ARG (-6)     /* bad argument (unsupported dimension, null pointer, bad handle) */
#define DH_ERR_QZERO (-7)   /* RuntimeError: Ellipsoid check failed q=0 (bounding.py:565-572) */
#define DH_ERR_NOMEM (-8)

/* likelihood / prior ids -- twins of dynesty_amd/problems.py */
#define DH_LIKE_GAUSS_IID 0  /* -0.5 sum v^2 + c                par = [c]        */
#define DH_LIKE_GAUSS_PREC 1 /* -0.5 v^T P v + c                par = [c, P]     */
#define DH_LIKE_EGGBOX 2     /* (2 + prod cos((2 tmax v - tmax)/2))^5  par = [tmax] */
#define DH_PRIOR_IDENTITY 0  /* v = u                                            */
#define DH_PRIOR_AFFINE 1    /* v = a (2u - 1) + b              par = [a, b]     */
#define DH_PRIOR_NORMAL 2    /* v = mu + sigma ndtri(u)         par = [mu, sigma]*/

/* per-dimension boundary flags (reference kwargs periodic / reflective,
 * utils.py:950-976 get_nonbounded) */
#define DH_BC_HARD 0
#define DH_BC_PERIODIC 1
#define DH_BC_REFLECT 2

typedef struct dh_ctx dh_ctx;

/* ---- context ----------------------------------------------------------- */
int dh_version(void);
int dh_device_count(void);
/* NULL on failure (no HIP device / bad ordinal): the message is in
 * dh_last_error(NULL).  There is no CPU fallback. */
dh_ctx* dh_create(int device);
void dh_destroy(dh_ctx* ctx);
const char* dh_last_error(dh_ctx* ctx);
int dh_sync(dh_ctx* ctx);
/* the hipStream_t every launch of this context goes to */
void* dh_stream(dh_ctx* ctx);

/* ---- device memory + events (for resident data and on-stream timing) --- */
void* dh_malloc(dh_ctx* ctx, uint64_t bytes);
int dh_free(dh_ctx* ctx, void* dptr);
int dh_memcpy_h2d(dh_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int dh_memcpy_d2h(dh_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);
int dh_memset(dh_ctx* ctx, void* dst_dev, int value, uint64_t bytes);
void* dh_event_create(dh_ctx* ctx);
int dh_event_destroy(dh_ctx* ctx, void* ev);
int dh_event_record(dh_ctx* ctx, void* ev);                     /* on dh_stream(ctx) */
int dh_event_elapsed_ms(dh_ctx* ctx, void* ev0, void* ev1, double* ms); /* syncs ev1 */

/* ---- problems: device twins of the user's loglikelihood/prior_transform --
 * callbacks (internal_samplers.py:957-958; dynesty.py:529-557).  Returns a
 * handle >= 0 or a DH_ERR_* code. */
int dh_problem_create(dh_ctx* ctx, int ndim, int like_id, const double* like_par,
                      int n_like_par, int prior_id, const double* prior_par,
                      int n_prior_par);
int dh_problem_destroy(dh_ctx* ctx, int problem);
/* logl/v for k points: the batched form of loglikelihood(prior_transform(u)) */
int dh_problem_eval(dh_ctx* ctx, int problem, int k, const double* u, double* v,
                    double* logl);

/* ---- RNG ---------------------------------------------------------------- */
/* numpy: [PCG64(c).state for c in SeedSequence(entropy).spawn(first+k)[first:]]
 * (utils.py:1002-1009 get_seed_sequence + :993-999 get_random_generator).
 * entropy_words = the entropy ints coerced to little-endian uint32 words. */
int dh_seed_children(dh_ctx* ctx, const uint32_t* entropy_words, int n_words,
                     uint32_t first_child, int k, uint64_t* states /* k*4 */);
/* test hook: from one state draw n_normal standard_normal() then n_unif
 * random(); returns the advanced state. */
int dh_rng_stream(dh_ctx* ctx, const uint64_t* state4, int n_normal, int n_unif,
                  double* normals, double* unifs, uint64_t* state4_out);

/* ---- membership ---------------------------------------------------------
 * MultiEllipsoid.contains/within/overlap (bounding.py:502-523) and
 * Ellipsoid.contains (bounding.py:286-305) for k points against m ellipsoids.
 *   mode 0: (x-c)^T A (x-c) < 1          (MultiEllipsoid, strict)
 *   mode 1: sqrt((x-c)^T A (x-c)) <= 1   (Ellipsoid)
 * count[i]  = number of ellipsoids containing x_i
 * mask      = bit matrix, word (a * ceil(k/64) + i/64), bit i%64 set iff x_i is
 *             inside ellipsoid a  (one 64-lane ballot per word); may be NULL
 * quad      = k*m quadratic forms (row i = point i); may be NULL */
int dh_contains(dh_ctx* ctx, const double* x, int k, int d, const double* ctrs,
                const double* ams, int m, int mode, int32_t* count,
                uint64_t* mask, double* quad);

/* ---- rebuild -------------------------------------------------------------
 * mode 0: MultiEllipsoid.update(points) without bootstrap (bounding.py:632-686)
 *         = bounding_ellipsoid (:1387-1461) + _bounding_ellipsoids (:1464-1563,
 *         incl. scipy kmeans2 k=2 iter=10) + improve_covar_mat (:1311-1384)
 *         + the all-points-covered check (:683-685).
 * mode 1: Ellipsoid.update(points) without bootstrap (bounding.py:345-375).
 * Outputs for the nells ellipsoids, in the reference's list order (left subtree
 * first): ctrs nells*d, covs/ams/axes nells*d*d (axes: column i = axis i, sorted
 * by ascending eigenvalue, sign fixed so the largest component is positive),
 * axlens nells*d, logvols nells; leaf_of_point[n] = index of the ellipsoid whose
 * cluster owns each point (may be NULL); nnodes = visited tree nodes (may be
 * NULL).  Errors: DH_ERR_VALUE / DH_ERR_CONTAIN / DH_ERR_REGION as the
 * reference raises; DH_ERR_NOMEM if more than max_ells ellipsoids result.
 * Dimensions: d <= 44 runs the LDS-resident kernel pipeline (rebuild.hip);
 * 44 < d <= 512 the wide path (wide.hip: multi-workgroup covariance /
 * eigensolver / Mahalanobis maximum; mode 0 as a host recursion over device node
 * work).  The batched / ragged forms below take any d <= 512 too; above 44 their
 * sets go through the wide constructions one after the other. */
int dh_rebuild(dh_ctx* ctx, const double* pts, int n, int d, int mode, int max_ells,
               int32_t* nells, double* ctrs, double* covs, double* ams, double* axes,
               double* axlens, double* logvols, int32_t* leaf_of_point,
               int32_t* nnodes);
/* `runs` independent live sets (pts: runs*n*d) in one launch, one workgroup per
 * run; outputs strided by max_ells per run; status[run] holds the per-run
 * DH_ERR code (the call itself only fails on launch errors). */
int dh_rebuild_batch_dev(dh_ctx* ctx, int runs, const double* pts, int n, int d,
                         int mode, int max_ells, int32_t* nells, int32_t* status,
                         double* ctrs, double* covs, double* ams, double* axes,
                         double* axlens, double* logvols, int32_t* leaf_of_point,
                         int32_t* nnodes);

/* Ragged batch: run r bounds the first n_arr[r] (<= n_max) rows of its
 * n_max x d block -- the B bootstrap replicas of _ellipsoid_bootstrap_expand
 * (bounding.py:1619-1648) in one launch. */
int dh_rebuild_ragged_dev(dh_ctx* ctx, int runs, const double* pts, int n_max,
                          const int32_t* n_arr, int d, int mode, int max_ells,
                          int32_t* nells, int32_t* status, double* ctrs, double* covs,
                          double* ams, double* axes, double* axlens, double* logvols);

/* Ellipsoid.__init__(ctr, cov) (bounding.py:201-240) for m covariance matrices:
 * eigen-decomposition -> axes (ascending, sign-canonical), axlens, am, logvol.
 * DH_ERR_VALUE if an eigenvalue is not positive/finite. */
int dh_ell_from_cov(dh_ctx* ctx, int m, int d, const double* covs, double* axes,
                    double* axlens, double* ams, double* logvols);
/* Ellipsoid.scale_to_logvol (bounding.py:242-276) applied in place to m
 * ellipsoids with per-ellipsoid targets (MultiEllipsoid.scale_to_logvol,
 * bounding.py:478-495, supplies logvol_ells + shift). */
int dh_scale_to_logvol(dh_ctx* ctx, int m, int d, double* covs, double* ams,
                       double* axes, double* axlens, double* logvols,
                       const double* targets);

/* improve_covar_mat (bounding.py:1311-1384; exercised by the reference's tests/test_ellipsoid.py:242-255
 * test_bounds): for each of m symmetric d x d matrices the 100-trial regularisation loop of the rebuild
 * kernels -- eigenvalue floor at 10 * max / 1e12 when the condition number exceeds 1e12, blend towards
 * the identity when the eigenvalues are not finite / not positive, identity after 100 failures.
 * Outputs: good[e] = 1 iff the input needed no change (the reference's `good_mat`), the returned
 * covariance, its inverse and the axes (eigenvectors * sqrt(eigenvalues), ascending, sign-canonical).
 * d <= 44. */
int dh_improve_covar_mat(dh_ctx* ctx, int m, int d, const double* covs_in, int32_t* good,
                         double* covs, double* ams, double* axes);

/* Sampler.update_bound's enlargement (sampler.py:506-508):
 * bound.scale_to_logvol(bound.logvol + log(enlarge)) for `runs` bounds laid out
 * as by dh_rebuild_batch_dev (max_ells slots per run, nells[run] live): every
 * live ellipsoid's target is its own logvol + log_enlarge (bounding.py:485-490). */
int dh_enlarge_batch_dev(dh_ctx* ctx, int runs, int max_ells, const int32_t* nells,
                         int d, double* covs, double* ams, double* axes,
                         double* axlens, double* logvols, double log_enlarge);

/* ---- proposals ----------------------------------------------------------
 * RWalkSampler.sample over a batch of k walkers = generic_random_walk +
 * propose_ball_point + randsphere (internal_samplers.py:866-1035,
 * bounding.py:1288-1297), prior/likelihood evaluated in-kernel.
 *   u0        k*ndim   start points (copies of live points)
 *   axes      m*ncdim*ncdim proposal frames, column i = axis i
 *   axes_idx  k        frame per walker (NULL: all use frame 0)
 *   bc        ndim     DH_BC_* flags or NULL (all hard)
 *   rng       k*4      PCG64 states in
 *   outputs   u,v k*ndim; logl k; naccept,nreject k; rng_out k*4 (may be NULL)
 * ncalls == walks for every walker (internal_samplers.py:925-944). */
int dh_rwalk_batch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim,
                   const double* u0, const double* axes, int m,
                   const int32_t* axes_idx, double scale, double loglstar,
                   int walks, const int8_t* bc, const uint64_t* rng, double* u,
                   double* v, double* logl, int32_t* naccept, int32_t* nreject,
                   uint64_t* rng_out);
int dh_rwalk_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim,
                       const double* u0, const double* axes, int m,
                       const int32_t* axes_idx, double scale, double loglstar,
                       int walks, const int8_t* bc, const uint64_t* rng,
                       double* u, double* v, double* logl, int32_t* naccept,
                       int32_t* nreject, uint64_t* rng_out);

/* Kernel form of the fused rwalk entry points (dh_rwalk_batch[_dev], dh_rwalk_batch_philox[_dev], the
 * rwalk stage of dh_ns_ensemble).  Two kernels implement generic_random_walk
 * (internal_samplers.py:866-986) on the same generator streams: one walker per lane (any ndim <= 32, any
 * options), and one walker on four lanes of a wavefront with the frame product / Gaussian quadratic form
 * on the fp64 matrix cores (built for ndim == ncdim in 2..32; periodic / reflective coordinates and every
 * fused prior included).  Accept / reject counts and generator end states of the two are identical,
 * coordinates agree to rounding (~1e-15: sums over a vector are taken in a different order).
 *   form 0 (default)  four lanes per walker wherever built: decided by the problem alone (its dimension, and
 *                     ndim == ncdim), never by the launch size or the device, so that a run's results do
 *                     not depend on how an ensemble is sharded
 *   form 1            one walker per lane always
 *   form 2            same as 0 (kept for callers of the round-3 interface)
 * The environment variable DH_RWALK_FORM sets the initial value.  DH_RWALK_ITEMS=0 keeps the PCG64 generator of
 * the four-lane form inside the walk kernel instead of a generator pass ahead of it (same streams, same results);
 * DH_RWALK_ITEMS_MB bounds that pass's buffer (default 1024: larger launches go in chunks of walkers). */
int dh_set_rwalk_form(dh_ctx* ctx, int form);

/* The four-lane form's PCG64 generator as a pass of its own ahead of the walk (on = 1, default: every walker's
 * walks x (ndim normals + 1 uniform) items written to a context buffer by one wavefront per walker, read back by
 * the walk kernel) or inside the walk kernel (on = 0).  Same streams, bit-identical results either way.
 * budget_bytes > 0 bounds the buffer (default 2^30): a launch whose streams need more runs in chunks of walkers;
 * 0 leaves the bound as it is.  (RWalkSampler.sample draws them in internal_samplers.py:1007-1021.) */
int dh_set_rwalk_items(dh_ctx* ctx, int on, long long budget_bytes);

/* Throughput mode of RWalkSampler.sample: the same walk (generic_random_walk, propose_ball_point,
 * randsphere; internal_samplers.py:866-1035, bounding.py:1288-1297) drawing from hiprand's Philox4x32-10
 * device generator instead of NumPy-compatible PCG64 streams: walker i uses subsequence sequence0 + i of
 * `seed`, starting `offset` draws in, so no generator state is read or written.  Normals are hiprand's
 * fp32 Box-Muller pairs widened to fp64 (the step direction is resolved to 6e-8; the proposal stays
 * exactly symmetric), uniforms hiprand_uniform_double.  Everything else -- frame product, wrap / reflect,
 * unitcheck, prior transform, log-likelihood, accept rule, counters -- is the parity kernel's code.
 * NOT stream-compatible with the reference: validated statistically (tests/test_gpu_philox.py: the
 * reference's KS tests of tests/test_ellipsoid.py on device output, chain statistics against the parity
 * mode).  Any ndim <= 512 (above 32 the wave-per-walker kernel).  A caller advances `offset` by at least walks * (ndim + 8) per launch (or changes
 * `seed`) to get fresh draws. */
int dh_rwalk_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                          const double* axes, int m, const int32_t* axes_idx, double scale,
                          double loglstar, int walks, const int8_t* bc, uint64_t seed,
                          uint64_t sequence0, uint64_t offset, double* u, double* v, double* logl,
                          int32_t* naccept, int32_t* nreject);
int dh_rwalk_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, const double* u0,
                              const double* axes, int m, const int32_t* axes_idx, double scale,
                              double loglstar, int walks, const int8_t* bc, uint64_t seed,
                              uint64_t sequence0, uint64_t offset, double* u, double* v, double* logl,
                              int32_t* naccept, int32_t* nreject);

/* Lock-step form for an arbitrary host likelihood: ONE propose_ball_point
 * (internal_samplers.py:989-1035) per walker -- draws, frame mat-vec,
 * wrap/reflect, unitcheck -- returning the proposals, their in-cube flags and the
 * advanced streams; the caller evaluates prior_transform/loglikelihood on the
 * host, applies the accept rule (internal_samplers.py:946-968) and calls again. */
int dh_rwalk_propose(dh_ctx* ctx, int k, int ndim, int ncdim, const double* u0,
                     const double* axes, int m, const int32_t* axes_idx, double scale,
                     const int8_t* bc, const uint64_t* rng, double* u_prop,
                     int32_t* inside, uint64_t* rng_out);

/* Slice sampling in lock step with a HOST likelihood.  generic_slice_step
 * (internal_samplers.py:1076-1206) evaluates F(x) = loglikelihood(prior_transform(u + x d)) at
 * every stepping-out / shrinking move; with an arbitrary Python likelihood the host runs that
 * state machine and this call supplies what each walker's random stream produces:
 *   kind 0  rslice: drhat = standard_normal(ndim) / |.|, dirs = axes . drhat * scale (:818-823)
 *   kind 1  slice : perm = the shuffled axis order of one sweep (:667-669)
 *   kind 2  nothing (advance / refill / commit only)
 * followed by a LOOKAHEAD of the next `nlook` Generator.random() values (rand0, doubling
 * draws, shrink draws) that is NOT committed to the stream.
 *   state6    k x 6 in/out: PCG64 state hi, lo, increment hi, lo, has_uint32, uinteger; on
 *             entry the stream is first advanced by consumed[i] draws (the part of the previous
 *             lookahead the host used; NULL = 0), on return it stands after the direction /
 *             shuffle draws and before the lookahead.
 *   axes      m x ndim x ndim (kind 0), axes_idx k or NULL
 *   dirs      k x ndim (kind 0); perm k x ndim (kind 1); look k x nlook */
int dh_slice_feed(dh_ctx* ctx, int k, int ndim, int kind, const double* axes, int m,
                  const int32_t* axes_idx, double scale, uint64_t* state6,
                  const int32_t* consumed, int nlook, double* dirs, int32_t* perm,
                  double* look);

/* RSliceSampler.sample (mode 0, internal_samplers.py:745-855) / SliceSampler.sample
 * (mode 1, :593-709) over k walkers; generic_slice_step + Neal's doubling
 * (:1038-1206) run as a per-lane state machine.  ncdim == ndim (dynesty.py:507-509).
 *   doubling   kwargs['slice_doubling'] on entry
 *   outputs    u,v k*ndim; logl, ncalls, nexpand, ncontract k;
 *              flags k: bit0 expansion_warning_set, bit1 x == 0 failure
 * The host-pointer form returns DH_ERR_SLICE if any walker failed. */
int dh_slice_batch(dh_ctx* ctx, int problem, int k, int ndim, int mode,
                   const double* u0, const double* axes, int m,
                   const int32_t* axes_idx, double scale, double loglstar,
                   int slices, int doubling, const uint64_t* rng, double* u,
                   double* v, double* logl, int32_t* ncalls, int32_t* nexpand,
                   int32_t* ncontract, int32_t* flags, uint64_t* rng_out);
int dh_slice_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int mode,
                       const double* u0, const double* axes, int m,
                       const int32_t* axes_idx, double scale, double loglstar,
                       int slices, int doubling, const uint64_t* rng, double* u,
                       double* v, double* logl, int32_t* ncalls, int32_t* nexpand,
                       int32_t* ncontract, int32_t* flags, uint64_t* rng_out);

/* Throughput mode of the slice samplers (see dh_rwalk_batch_philox): the same kernels drawing from
 * hiprand's Philox4x32-10 generator -- directions (internal_samplers.py:820) from fp32 Box-Muller normals
 * widened to fp64, the axis shuffle (:673) by masked rejection on 32-bit draws, rand0 / doubling coin /
 * shrink uniforms (:1099, 1151, 1173) from 53-bit doubles in [0, 1) -- walker i on subsequence
 * sequence0 + i of `seed`, `offset` draws in.  Consumption depends on the data: advance `offset` by more
 * than a walker can draw in one call (2^24 is what dh_ns_ensemble uses).  Any ndim <= 512 (above 32, and
 * at dimensions without a register instantiation, the wave-per-walker kernels of the wide path). */
int dh_slice_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                          const double* axes, int m, const int32_t* axes_idx, double scale,
                          double loglstar, int slices, int doubling, uint64_t seed, uint64_t sequence0,
                          uint64_t offset, double* u, double* v, double* logl, int32_t* ncalls,
                          int32_t* nexpand, int32_t* ncontract, int32_t* flags);
int dh_slice_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int mode, const double* u0,
                              const double* axes, int m, const int32_t* axes_idx, double scale,
                              double loglstar, int slices, int doubling, uint64_t seed, uint64_t sequence0,
                              uint64_t offset, double* u, double* v, double* logl, int32_t* ncalls,
                              int32_t* nexpand, int32_t* ncontract, int32_t* flags);

/* UniformBoundSampler.sample (internal_samplers.py:243-340) with the bound's
 * sample() inlined: m == 1 Ellipsoid.sample (bounding.py:307-319), m > 1
 * MultiEllipsoid.sample incl. the 1/q overlap rejection (bounding.py:525-590),
 * m == 0 UnitCubeSampler.sample (internal_samplers.py:364-441).
 *   ctrs m*ncdim, axes/ams m*ncdim*ncdim, cumprob m = cumsum(exp(logvol_ells -
 *   logvol)) (rand_choice, bounding.py:1300-1308); bc: DH_BC_* per dim or NULL
 *   max_tries: per-walker guard (<= 0: 2^32 tries, then DH_ERR with "exceeded max_tries"); ncalls = likelihood calls. */
int dh_unif_batch(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m,
                  const double* ctrs, const double* axes, const double* ams,
                  const double* cumprob, double loglstar, const int8_t* bc,
                  const uint64_t* rng, int64_t max_tries, double* u, double* v,
                  double* logl, int32_t* ncalls, uint64_t* rng_out);
int dh_unif_batch_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m,
                      const double* ctrs, const double* axes, const double* ams,
                      const double* cumprob, double loglstar, const int8_t* bc,
                      const uint64_t* rng, int64_t max_tries, double* u, double* v,
                      double* logl, int32_t* ncalls, int32_t* flags,
                      uint64_t* rng_out);

/* Throughput mode of UniformBoundSampler.sample / UnitCubeSampler.sample (m = 0: north_star's "hiprand for the
 * unit-cube draws"): ellipsoid choice, randsphere (bounding.py:1288-1297, 543-590), the 1/q test and the
 * unit-cube uniforms (internal_samplers.py:428) from hiprand's Philox4x32-10 generator, keyed as in
 * dh_rwalk_batch_philox.  Consumption depends on the data: advance `offset` generously between calls. */
int dh_unif_batch_philox(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                         const double* axes, const double* ams, const double* cumprob, double loglstar,
                         const int8_t* bc, uint64_t seed, uint64_t sequence0, uint64_t offset,
                         int64_t max_tries, double* u, double* v, double* logl, int32_t* ncalls);
int dh_unif_batch_philox_dev(dh_ctx* ctx, int problem, int k, int ndim, int ncdim, int m, const double* ctrs,
                             const double* axes, const double* ams, const double* cumprob, double loglstar,
                             const int8_t* bc, uint64_t seed, uint64_t sequence0, uint64_t offset,
                             int64_t max_tries, double* u, double* v, double* logl, int32_t* ncalls,
                             int32_t* flags);

/* Bound.sample / samples from ONE generator: Ellipsoid.sample(s)
 * (bounding.py:307-334) for m == 1, MultiEllipsoid.sample(s)
 * (bounding.py:525-606) for m > 1 (return_q != 0: no internal 1/q rejection).
 * xs nsamp*d, idxs/qs nsamp, state4_out = the generator after the draws. */
int dh_bound_draw(dh_ctx* ctx, const uint64_t* state4, int nsamp, int d, int m,
                  const double* ctrs, const double* axes, const double* ams,
                  const double* cumprob, int return_q, double* xs, int32_t* idxs,
                  int32_t* qs, uint64_t* state4_out);

/* One queue consumption of the static run loop for `runs` independent runs -- the iteration loop of
 * Sampler.sample over ONE queue fill (sampler.py:1070-1185) with _new_point's queue rule
 * (sampler.py:741-776): while the dlogz criterion has not fired, the worst live point dies and is replaced
 * by the next queue entry whose logl beats it (entries that do not are discarded, their calls still
 * charged); every death takes the volume step ln((N+1)/N) (inside a likelihood plateau: the plateau's step) and one step of progress_integration
 * (utils.py:1470-1492).  This is the `ns_consume` stage of dh_ns_ensemble as an operator of its own
 * (used to hold it to the oracle's restatement of the reference loop; a host-driven loop can use it too).
 *   live_logl  runs x nlive   in/out, slot order
 *   q_logl     runs x K       the queue's log-likelihoods in queue order; q_ncalls their call counts
 *   state      runs x 8       in/out: logvol, logz, h, logzvar, loglstar of the last dead point
 *                             (-1e300 before the first), it, ncall, [out only] the current worst logl
 *   dead_logl / dead_slot / dead_src   runs x K: this call's deaths in order (logl, slot, queue index of
 *                             the replacement); ndead[r] of them are valid; stopped[r] = dlogz fired
 *   live_it / dead_it / dead_nc  optional (NULL together): the per-point bookkeeping of saved_run
 *                             (sampler.py:1107, 1141, 1165-1182).  live_it runs x nlive in/out = iteration at
 *                             which the point living in each slot was proposed (0 = initial point; the
 *                             reference counts iterations from 1); dead_it runs x K = that value for every
 *                             dead point ('it'); dead_nc runs x K = likelihood calls spent on its replacement
 *                             ('nc': every queue entry popped since the previous death); dead_slot is 'id'.
 *   plateau    optional, runs x 2 in/out: the likelihood-plateau mode of the reference (sampler.py:1112-1127,
 *                             1190-1193) carried between calls -- deaths still to be taken with the plateau's
 *                             constant volume step (0 = off) and the logarithm of that step's volume; NULL:
 *                             the call starts outside plateau mode and the state is not handed back
 * Equal log-likelihoods die lowest slot first, as in the reference (np.argmin, sampler.py:1107), and when the
 * worst point shares its value with others the deaths take the reference's plateau volume steps. */
int dh_ns_consume(dh_ctx* ctx, int runs, int nlive, int queue_size, double dlogz, double* live_logl,
                  const double* q_logl, const int32_t* q_ncalls, double* state, double* dead_logl,
                  int32_t* dead_slot, int32_t* dead_src, int32_t* ndead, int32_t* stopped, int32_t* live_it,
                  int32_t* dead_it, int32_t* dead_nc, double* plateau);

/* ---- device-resident ensemble of static nested-sampling runs (BASELINE config
 * C5): the loop of Sampler.sample (sampler.py:932-1212) with a
 * proposal queue of `queue_size` rwalk / rslice / slice walkers per run
 * (sampler.py:676-778; tune / tune_slice once per fill),
 * unit-cube start, MultiEllipsoid (bound_multi=1) or Ellipsoid bound rebuilt
 * every walks*nlive calls and enlarged by `enlarge`, RWalkSampler.tune, evidence
 * integration (utils.py:1470-1492) and the final live points -- all on the
 * device for `runs` independent runs at once.  Dimensions: any ndim <= 512 (above 32 the walkers are the
 * wave-per-walker kernels, above 44 the bound is the multi-workgroup Ellipsoid.update: BASELINE config C4 -- or,
 * bound_multi=1, the wide MultiEllipsoid.update, whose recursion the host drives: such a run synchronises the
 * stream on its rebuild fills); sample='unif' ndim <= 32.  Sizes: nlive <= 65535 (slots travel as 16-bit indices),
 * queue_size <= 2048.  The queue consumption keeps a run's keys, their sorted order and the queue in LDS while
 * 12 nlive + 2 P + 52 queue_size bytes <= 150 KB (P = nlive rounded up to a power of two) and nlive <= 8192; beyond
 * that it selects the queue_size + 1 smallest live points from the keys in global memory first -- the only ones a fill
 * can touch -- and works on those (same deaths, same evidence).  DH_ERR_ARG otherwise.  Run r seeds from
 * SeedSequence(entropy) children keyed on first_run + r (independent of how the
 * ensemble is sharded).  records: runs x 8 doubles {logz, logzerr, niter, ncall,
 * h, nbound, status (0 ok, 1 max_fills hit, -1 failed), eff%}.
 * dead_logl_out (optional): runs x max_iter dead-point log-likelihoods;
 * live_logl_out (optional): runs x nlive log-likelihoods of the final live points
 * (together they are what utils.merge_runs needs to combine the ensemble);
 * dead_u_out (optional): runs x max_iter x ndim unit-cube coordinates of the dead
 * points in death order (only the first niter rows of a run are written);
 * live_u_out (optional): runs x nlive x ndim final live points (slot order, matching
 * live_logl_out) -- the posterior samples of the runs.
 * dead_id_out / dead_it_out / dead_nc_out (optional, NULL together with live_it_out): runs x max_iter
 * int32 = the reference's per-point 'id' (live slot), 'it' (iteration at which the point was proposed,
 * counted from 1; 0 = initial point) and 'nc' (likelihood calls spent on the point's replacement) of every
 * dead point (sampler.py:1165-1182) -- what utils.merge_runs carries through as samples_id / samples_it /
 * ncall; live_it_out: runs x nlive = 'it' of the final live points (slot order; their id is the slot,
 * their nc is 1 by the reference's convention, sampler.py:880-886).
 * rebuild_sync = 0 keeps the reference's schedule per run (rebuild when ncall has
 * advanced by update_interval, sampler.py:625-674); rebuild_sync = 1 lets every run
 * that already has a bound rebuild whenever ANY run of the ensemble is due (its
 * rebuild comes early, never late): a rebuild is a latency-bound tree construction
 * that costs about the same for one run or the whole ensemble.
 * sampler 6 (PCG64) / 7 (Philox): the uniform sampler inside the run's bound (UniformBoundSampler.sample,
 * internal_samplers.py:243-340; rand_choice over the ellipsoids, randsphere, 1/q acceptance), ndim <= 32; the
 * bound is rebuilt every nlive calls (its update_bound_interval_ratio is the base class's 1,
 * internal_samplers.py:88-94), `walks` is ignored, a walker whose draw no ellipsoid holds fails its run as the
 * reference's RuntimeError does.  bootstrap = B > 1: every rebuild is followed by the bootstrap expansion of
 * bounding.py:381-400 / 688-703 -- B resampled replicas per run (bounding.py:1593-1648), all rebuilt in one ragged
 * batch, the bound scaled by max(1, largest normalised distance of a left-out point)^ndim; the reference's default
 * for sample='unif' is (enlarge 1, bootstrap 5), dynesty.py:169-200.  (Above ndim = 44 the replicas are built one
 * by one by the wide constructions, with the host between them.)  0: none.
 * rebuild_every = n: bounds are built only every n-th queue fill; a run that becomes due in between idles (proposes
 * and consumes nothing) until then.  Runs are independent, so a run's own sequence -- and with PCG64 streams its
 * result, bit for bit -- is that of the reference schedule (n = 1).  Once n reaches the number of fills a run needs
 * to spend its update interval every run is due by the next rebuild fill: the ensemble rebuilds together, one
 * latency chain per interval instead of one per fill (64 C2 runs: 0.335 s at n = 1, 0.20 s at n >= 4).  0: chosen
 * from (sampler, nlive, queue_size) alone: ceil(nlive / K) for rwalk, ceil(1.3 nlive / (4.4 K)) for the slice
 * samplers, ceil(1.3 nlive / (1.7 K)) for unif; at most 16.
 * An rwalk walker that accepted no step returns its start point with that live point's own stored ln L (an exact tie, as
 * the reference's re-evaluation gives: internal_samplers.py:970-975, sampler.py:1107-1119).
 * max_fills (0: 10^6) bounds the fills the ENSEMBLE is taken through, idle ones included: a fill in which a run
 * waits for its rebuild counts for that run as well, so with rebuild_every > 1 a run reaches status 1 after fewer of
 * its own fills than max_fills (n_fills_out is the same count). */
int dh_ns_ensemble(dh_ctx* ctx, int problem, int runs, int nlive, int ndim,
                   int queue_size, int sampler /* 0 rwalk, 1 rslice, 2 slice; + 3: unit-cube phase and proposals from hiprand Philox streams; 6 / 7 unif */,
                   int walks /* or slices */, int bound_multi,
                   int rebuild_sync /* 1: all runs rebuild together, see below */, double dlogz,
                   double enlarge, int64_t max_fills, int64_t max_iter,
                   const uint32_t* entropy_words, int n_words, uint32_t first_run,
                   double* records, double* dead_logl_out, double* live_logl_out,
                   double* dead_u_out, double* live_u_out, int64_t* n_fills_out, int32_t* dead_id_out,
                   int32_t* dead_it_out, int32_t* dead_nc_out, int32_t* live_it_out, int bootstrap,
                   int rebuild_every /* bounds are built every n-th fill, runs due in between wait; 0: chosen from the shape */);

/* Options of the resident run loop that the reference takes in NestedSampler(...) / run_nested(...) and that
 * dh_ns_ensemble otherwise fixes at the reference's defaults.  They are context state: set before a
 * dh_ns_ensemble call, they hold until set again; value NaN restores the default.
 *   DH_NS_OPT_UPDATE_INTERVAL   calls between bound updates (dynesty.py:213-234 `update_interval` as a number of calls;
 *                               a caller holding the ratio form multiplies by nlive); default: the sampler's ratio
 *                               (1 unif, walks rwalk, slices rslice, slices * ndim slice) * nlive
 *   DH_NS_OPT_FIRST_MIN_NCALL   first_update['min_ncall'] (sampler.py:625-674), default 2 nlive
 *   DH_NS_OPT_FIRST_MIN_EFF     first_update['min_eff'] in per cent, default 10
 *   DH_NS_OPT_MAXITER           run_nested(maxiter): the loop stops once its counter exceeds it, i.e. after maxiter + 1
 *                               deaths (sampler.py:1076-1083); default none
 *   DH_NS_OPT_MAXCALL           run_nested(maxcall): stops once the likelihood calls (initial points included)
 *                               exceed it; default none
 *   DH_NS_OPT_LOGL_MAX          run_nested(logl_max): stops once the last dead point's ln L exceeds it; default +inf
 *   DH_NS_OPT_ADD_LIVE          run_nested(add_live): 0 = the record is the dead points' running evidence, the
 *                               final live points stay out (sampler.py:1319-1341); default 1
 *   DH_NS_OPT_FORCED_EXACT      1 (the default since round 5) = the reference's protocol: Sampler.propose_live's forced
 *                               bound update (sampler.py:484-489) belongs to the fill that finds a start point outside
 *                               the bound -- the queue entries up to and including the first one outside keep their
 *                               axes from the old bound, the later ones take theirs from the new one -- and the regular
 *                               bound is built as the reference builds it, without the point that the previous fill's
 *                               last queue entry brought in (update_bound_if_needed runs before that replacement:
 *                               sampler.py:771-772, 1176-1185).  Scheduling: the forced update is built together with
 *                               the regular updates of the next fill that builds bounds (one masked rebuild sequence
 *                               for both); until then the flagged run keeps its queue and sits the fills out, which
 *                               leaves its own sequence of events unchanged (runs are independent).
 *                               0 = the late form: the run is flagged, walks this fill from the old bound and rebuilds
 *                               before its next fill; the regular bound includes the newest point (9-20 % fewer bound
 *                               updates than the reference; a few per cent faster).  The uniform sampler has no start
 *                               points: the option does not apply.  1 switches DH_NS_OVERLAP off.
 * A run ended by maxiter / maxcall / logl_max ends normally (status 0), as the reference's does.
 * Options (and dh_ns_set_boundary's flags) apply to the NEXT dh_ns_ensemble call only: that call takes them and the
 * context forgets them, however the call ends. */
enum {
  DH_NS_OPT_UPDATE_INTERVAL = 0,
  DH_NS_OPT_FIRST_MIN_NCALL = 1,
  DH_NS_OPT_FIRST_MIN_EFF = 2,
  DH_NS_OPT_MAXITER = 3,
  DH_NS_OPT_MAXCALL = 4,
  DH_NS_OPT_LOGL_MAX = 5,
  DH_NS_OPT_ADD_LIVE = 6,
  DH_NS_OPT_FORCED_EXACT = 7,
  DH_NS_OPT_COUNT = 8
};
int dh_ns_set_option(dh_ctx* ctx, int key, double value);

/* NestedSampler(periodic=, reflective=) for the resident loop (dynesty.py:126-142: the lists only reach the internal
 * sampler): bc = ndim DH_BC_* flags, or NULL / ndim <= 0 for none.  Taken by the next dh_ns_ensemble call (one-shot, like the options; a rejected call leaves installed flags as they were);
 * dh_ns_ensemble refuses a flag array whose length is not its ndim.  rwalk wraps / reflects the flagged coordinates and
 * tests them against (-0.5, 1.5) (internal_samplers.py:1023-1032), the uniform sampler only widens its unitcheck
 * (:301-314); the slice samplers ignore the flags, as the reference's do (its `nonperiodic` kwarg is never set). */
int dh_ns_set_boundary(dh_ctx* ctx, int ndim, const int8_t* bc);

/* The bootstrap expansion factor on its own (host pointers; what the resident loop runs after a rebuild):
 * for each of `runs` point sets (runs x n x d) the max over `bootstrap` replicas of max(1, largest normalised
 * distance of a left-out point to the replica's Ellipsoid (multi = 0) / nearest of its MultiEllipsoid's ellipsoids
 * (multi = 1)) -- _ellipsoid_bootstrap_expand, bounding.py:1619-1648.  Replica b of run r resamples with the
 * PCG64 stream seeded (ent[4r], ent[4r+1] + b) with increment words (ent[4r+2], ent[4r+3] + 2b)
 * (oracle/nested_ref.py boot_generator).  expand: runs doubles; n_in (optional): runs x bootstrap sample sizes
 * (distinct points drawn, after _bootstrap_points' repairs). */
int dh_bootstrap_expand(dh_ctx* ctx, int runs, const double* pts, int n, int d, int multi, int bootstrap,
                        const uint64_t* ent, double* expand, int32_t* n_in);

/* ---- RadFriends / SupFriends (bounding.py:734-1263, 1651-1702) ----------------
 * kind: 0 = 'balls' (RadFriends, Euclidean norm), 1 = 'cubes' (SupFriends, max norm).
 * All matrices are d x d row-major host arrays; axes = sqrtm(cov) and axes_inv = pinvh(axes)
 * are symmetric, am = pinvh(cov).
 *
 * dh_friends_update = RadFriends.update / SupFriends.update (bounding.py:876-957, 1141-1222):
 * covariance of the points after re-centring every single-linkage cluster (cut at Mahalanobis
 * distance 1 in the PREVIOUS metric am_prev, evaluated with the arithmetic of scipy's pdist so
 * that the knife-edge pair -- the radius puts the loneliest point exactly at distance 1 -- falls
 * the reference's way; NULL = use_clustering=False), its symmetric square root and pseudo-inverses, the points in
 * the whitened frame, and the radius / half-side = max nearest-neighbour distance: leave-one-out
 * (nboot = 0) or, per bootstrap replica b, from the left-out points to the resampled ones
 * (in_mask[b*n + i] != 0 iff point i was resampled; the index bookkeeping of
 * _bootstrap_points, bounding.py:1593-1616, stays on the host).  Outputs are already scaled by
 * the radius (cov r^2, am / r^2, axes r, axes_inv / r); logvol = ln volume of ONE shape.
 * DH_ERR_VALUE: non-finite or singular covariance, or zero radius (the reference divides by 0). */
int dh_friends_update(dh_ctx* ctx, const double* pts, int n, int d, int kind,
                      const double* am_prev, int nboot, const uint8_t* in_mask,
                      double* cov, double* am, double* axes, double* axes_inv,
                      double* logvol, double* rmax, int32_t* nclusters);

/* RadFriends.within / overlap / contains for m candidate points (bounding.py:777-793,
 * 1043-1064): counts[c] = number of balls / cubes containing x_c; bits (optional,
 * m x ceil(n/64) words) = which ones (bit j of word j/64). */
int dh_friends_within(dh_ctx* ctx, const double* ctrs, int n, int d, int kind,
                      const double* axes_inv, const double* x, int m, int32_t* counts,
                      uint64_t* bits);

/* RadFriends.sample(s) / SupFriends.sample(s) from ONE generator (bounding.py:795-847,
 * 1066-1117), same draw order as the reference (balls: d normals + 1 uniform; cubes: d uniforms;
 * then integers(n) on the buffered 32-bit stream when n > 1; then 1 uniform iff q > 1 and not
 * return_q).  state6 = {state hi, state lo, inc hi, inc lo, has_uint32, uinteger}. */
int dh_friends_draw(dh_ctx* ctx, const uint64_t* state6, int nsamp, const double* ctrs, int n,
                    int d, int kind, const double* axes, const double* axes_inv, int return_q,
                    double* xs, int32_t* qs, uint64_t* state6_out);

/* UniformBoundSampler.sample over a queue of k walkers with a RadFriends / SupFriends bound
 * (internal_samplers.py:243-340 with bounding.py:795-831 / 1066-1101 as bound.samples(1)):
 * per try one draw from the union of shapes (1/q rule, brute-force overlap over the n centres),
 * unitcheck, prior transform and likelihood of the device problem, until logl > loglstar.
 * Same arguments and outputs as dh_unif_batch; ncdim == ndim (the reference's friends bounds
 * need it too: their centres are the full live points).
 * problem = -1 (also accepted by dh_unif_batch) selects the lock-step form for an arbitrary
 * host likelihood: every walker returns the next candidate of its stream that lies in the bound
 * and passes unitcheck (u, rng_out; v / logl / ncalls may be NULL) and the caller evaluates it.
 * rng32 / rng32_out (optional, k x 2 {has_uint32, uinteger}) carry NumPy's buffered 32-bit half,
 * which integers(n) consumes, from one lock-step round to the next. */
int dh_unif_friends_batch(dh_ctx* ctx, int problem, int k, int ndim, int kind, const double* ctrs,
                          int n, const double* axes, const double* axes_inv, double loglstar,
                          const int8_t* bc, const uint64_t* rng, int64_t max_tries, double* u,
                          double* v, double* logl, int32_t* ncalls, uint64_t* rng_out,
                          const uint64_t* rng32, uint64_t* rng32_out);

#ifdef __cplusplus
}
#endif
#endif /* DYNHIP_H */
