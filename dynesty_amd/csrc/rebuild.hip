// MultiEllipsoid / Ellipsoid rebuild on the device (kernels K1-K4 of SURVEY.md).
//
// One workgroup owns one live set ("run") and executes the whole rebuild --
// bounding ellipsoid of the root, the recursive k=2 split
// (reference bounding.py:1464-1563) as a level-ordered worklist, the bottom-up
// BIC-style accept test, and the coverage check of MultiEllipsoid.update
// (bounding.py:632-686) -- in ONE launch: no host round trips, no inter-
// workgroup synchronisation.  An ensemble of runs is a grid of workgroups.
//
// Data layout: points stay row-major (N x D, fp64) in HBM/L2; a permutation
// array keeps every tree node a contiguous segment (stable partition = the
// order of `points[labels == k]`).  Node tiles of TP points are gathered
// (coalesced along D) into LDS, centred or rescaled on the way in; covariance
// entries / centroid sums are accumulated from LDS by a fixed thread->entry
// map so results are deterministic.  D x D eigenproblems are solved by one
// wavefront with a parallel-order cyclic Jacobi in LDS.
#include <math.h>
#include <stdlib.h>

#include "ctx.h"
#include <type_traits>
#include <vector>
#include "eig_wave.h"

using namespace dh;
using dh_eig::jacobi_wave;
using dh_eig::sort_eigs_wave;
using dh_eig::wave_sync;

namespace {

constexpr int kThreads = 256;
constexpr double kRoundDelta = 1e-3;  // bounding.py:1420
constexpr double kMaxCond = 1e12;     // bounding.py:1311
constexpr double kEigMult = 10.0;     // bounding.py:1326
constexpr int kNTries = 100;          // bounding.py:1311

// per-node record in global scratch
struct Node {
  int start, count;
  int parent, child0, child1;
  int depth;
  int split;        // 1 if children were created
  int res_start, res_len;  // result list (indices of nodes) after the accept test
  int fast;         // 1: the record holds the eigen-free form (spd_fast): am, major axis, logvol only
  int has_mean;     // 1: the record's centre slot already holds the node's mean (from the k-means cluster sums)
  double logvol;
  double fmax;      // max_i delta_i^T am delta_i over the node's own points, with the stored am (after the rescale)
};

struct RebuildArgs {
  const double* pts;  // runs x n x d
  int n, d, runs;
  int mode;           // 0: MultiEllipsoid.update, 1: Ellipsoid.update (no split)
  int max_nodes, max_ells;
  double prefactor;   // logvol_prefactor(d), bounding.py:1271-1285
  // per-run scratch (strided by run)
  int* perm;          // runs x n
  int* perm2;         // runs x n
  unsigned char* lab; // runs x n
  Node* nodes;        // runs x max_nodes
  double* estore;     // runs x max_nodes x ES   (ctr D | cov D^2 | am D^2 | axes D^2 | axlens D)
  int* reslist;       // runs x (max_nodes * 2 + ...) result-list arena
  int reslist_cap;
  // outputs (strided by run)
  int* nells;         // runs
  int* status;        // runs
  double* ctrs;       // runs x max_ells x D
  double* covs;       // runs x max_ells x D^2
  double* ams;
  double* axes;
  double* axlens;     // runs x max_ells x D
  double* logvols;    // runs x max_ells
  int* leaf_of_point; // runs x n (index into the output list) or null
  int* nnodes_out;    // runs or null
  // level pipeline
  int maxw;           // max splittable nodes of one run at one level: n / (4d) + 1
  int levels;         // number of (k_split, k_ell) level steps launched
  int* nnodes_dev;    // runs
  int* nsplit;        // (levels+1) x runs
  int* nell;          // levels x runs
  int* split_list;    // 2 x runs x maxw   (by level parity)
  int* ell_list;      // levels x runs x 2 maxw (one list per level)
  // k-means parts: a splittable node of c points is worked on by ceil(c / TP) workgroups
  // ("parts"), each keeping its TP points resident in LDS for all ten iterations
  int fin_extra_off;  // k_finish: byte offset of the LDS node/result-list copies (0: keep them in global memory)
  int fin_res_lds;    // 1: the result-list arena is in LDS too
  int tps;            // points per k-means part (the tile k_split keeps resident); a node of c points has ceil(c / tps) parts
  int maxp;           // max parts of one run at one level: n / tps + maxw + 1
  int* nparts;        // (levels+1) x runs
  int* part_list;     // 2 x runs x maxp x 2   (by level parity): (slot in split_list, part index)
  int* part_base;     // 2 x runs x maxw       first part slot of the node in split_list slot
  int* kbar;          // levels x runs x maxw  arrive counters of the part barriers
  int* kerr;          // runs: error raised inside k_split (folded into status by the next kernel)
  double* fin_lse;    // runs x max_nodes: k_finish scratch (logsumexp of a node's result list)
  int* fin_int;       // runs x max_nodes x 2: k_finish scratch (accepted split / on the output path)
  int* rbar;          // runs x kBarStride: barrier counters of the cooperative root
  double* rootbuf;    // runs x rootbuf_stride: partials exchanged by the root's parts
  size_t rootbuf_stride;
  double* kpart;      // 2 x runs x maxp x (2d + 2): per-part partial sums, by iteration parity
  double* scale_g;    // runs x d
  double* pts_scaled; // runs x n x d : points / root std, written once by k_root (k-means input)
  const int* active;  // runs or null: only runs with active[run] != 0 are rebuilt
  const int* n_arr;   // runs or null: per-run point count (<= n; rows beyond it are padding)
  int fast;           // 1: tree nodes take the eigen-free path (spd_fast); k_out_eig solves the outputs
  double* root_eig;   // runs x (2 D^2 + D + 2): am | axes | axlens | logvol | ok -- the root's eigen-system (k_root_eig)
  int* out_node;      // runs x max_ells: node behind output ellipsoid m (k_finish -> k_out_eig)
  int* out_fast;      // runs x max_ells: 1 = that node's record is the eigen-free form
  // persistent work-queue form of the tree (k_tree): no levels -- a node's split is queued when its
  // ellipsoid exists, its children's ellipsoids when its last part has finished the partition
  int tree;           // 1: queue_split / split_body / ell_body feed the queue instead of the level lists
  int tree_from;      // level pipeline: splits for levels >= tree_from are queued for the k_tree tail instead
  unsigned long long* tq_items;  // tq_cap work items, 0 = not published yet
  int* tq_ctl;        // [0] head (next ticket), [16] tail (next free slot), [32] items queued or in flight, [48] error
  int tq_cap;
  int* nbar;          // runs x max_nodes x kBarStride: per node [0] part barrier, [1] parts done, [2] first k-means partial slot
  int* kp_top;        // runs: partial-sum slots handed out by the queue form (multi-part nodes only; capacity kp_cap per run)
  int kp_cap;
  int tq_sleep, tq_nocoh;  // diagnostics (DH_TREE_SLEEP, DH_TREE_NOCOH)
  unsigned epoch;          // rebuild launches of this context so far: part of the tag of the k-means partials
  int root_run0;           // k_root_parts: first run of this launch (the runs go in chunks that are co-resident)
};

#ifdef DH_REBUILD_TIMING
#ifndef DH_PHASE_WG
#define DH_PHASE_WG 0  // the workgroup of every kernel whose phases are summed (-DDH_PHASE_WG=k: another one)
#endif
__device__ long long g_phase_cycles[16];
// (round 6) ... and per level of k_ell: g_ell_cycles[level][phase], the level filed by the kernel in g_ph_level
__device__ long long g_ell_cycles[8][16];
__shared__ int g_ph_level;
#define PH_T0() long long t0_ = clock64()
#define PH_ADD(i)                                                      \
  do {                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == DH_PHASE_WG) {               \
      long long t1_ = clock64();                                       \
      g_phase_cycles[i] += t1_ - t0_;                                  \
      if (g_ph_level >= 0 && g_ph_level < 8) g_ell_cycles[g_ph_level][i] += t1_ - t0_; \
      t0_ = t1_;                                                       \
    }                                                                  \
  } while (0)
#define PH_LEVEL(l)                          \
  do {                                       \
    if (threadIdx.x == 0) g_ph_level = (l);  \
    __syncthreads();                         \
  } while (0)
// per level: phases of k_split's workgroup DH_PHASE_WG (0 prologue + staging, 1 Lloyd iterations, 2 partition, 3 child
// records, 4 number of iterations, 5 parts of the node, 6 points of the node)
__device__ long long g_lvl_cycles[16][16];
#define LV_T0() long long lt0_ = clock64()
#define LV_ADD(lvl, i)                                                 \
  do {                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == DH_PHASE_WG && (lvl) < 16) { \
      long long lt1_ = clock64();                                      \
      g_lvl_cycles[lvl][i] += lt1_ - lt0_;                             \
      lt0_ = lt1_;                                                     \
    }                                                                  \
  } while (0)
#define LV_SET(lvl, i, v)                                              \
  do {                                                                 \
    if (threadIdx.x == 0 && blockIdx.x == DH_PHASE_WG && (lvl) < 16) g_lvl_cycles[lvl][i] += (v); \
  } while (0)
// per workgroup of the level kernels: wall clock (100 MHz) at entry and exit, [kernel 0 = k_split, 1 = k_ell][level][workgroup]
constexpr int kWgMax = 4096;
__device__ long long g_wg_clock[2][8][kWgMax][2];
#define WG_STAMP(kern, lvl, which)                                                              \
  do {                                                                                          \
    if (threadIdx.x == 0 && (lvl) < 8 && blockIdx.x < kWgMax) g_wg_clock[kern][lvl][blockIdx.x][which] = wall_clock64(); \
  } while (0)
#else
#define PH_T0()
#define PH_ADD(i)
#define PH_LEVEL(l)
#define LV_T0()
#define WG_STAMP(kern, lvl, which)
#define LV_ADD(lvl, i)
#define LV_SET(lvl, i, v)
#endif

// ---- LDS carve-up ----------------------------------------------------------
struct Lds {
  double* tile;   // TP x LD
  double* A;      // D x LD   (cov / Jacobi work)
  double* V;      // D x LD
  double* AM;     // D x LD
  double* AX;     // D x LD   (axes)
  double* mean;   // D
  double* scale;  // D
  double* lam;    // D
  double* cen;    // 2 x D
  double* sums;   // 2 x D + 2
  double* red;    // kThreads
  double* rc;     // 64 (rotation c)
  double* rs;     // 64
  double* kred;   // 4 waves x 2 clusters x 48: k-means partial sums (overlays red | rc | rs)
  int* ri;        // 256 ints (pair indices, scan scratch, ...)
  int* ibuf;      // staging: a tile's permutation indices (overlays red | rc | rs: free while a tile is gathered)
  int ibuf_cap;   // ints that fit there
  int* perm_sort; // D
  int TP, LD, DP, DPlog;
  int KG;         // waves per partial-sum group of the k-means (the whole workgroup unless a bigger tile stands in for parts)
  double* JA[2];  // Jacobi work buffers, P x JLD (P = D rounded up to even)
  double* JV[2];
  int JLD;
  bool j_alias;   // true: JA/JV overlay the tile (D too large for separate buffers)
  // what the tile currently holds (see stage_tile)
  mutable const double* c_pts;
  mutable int c_start, c_cnt, c_how;
  // k_tree: producer and consumer of a node's permutation range, record and tree entry are different
  // workgroups of the SAME kernel (other CUs, other XCDs with their own L2): everything that crosses goes
  // through agent-scope stores / loads (write-through / bypassing), see parts_barrier
  bool coh;
};

// ---- lane exchanges inside a row of 16 on the vector pipe ---------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 (two per double, an LDS round trip of ~130 cycles each, and every reduction
// step waits for the one before).  For the partner masks 1, 2, 4 and 8 the same exchange is a DPP move: quad_perm for
// 1 and 2, row_shl:4 / row_shr:4 under complementary bank masks for 4, row_ror:8 for 8; 16 and 32 cross rows:
// gfx950's v_permlane16_swap / v_permlane32_swap and a select.  Same partner, same operands: the same bits as the shuffle.
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_mov_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK, false);
}
template <int S>
__device__ __forceinline__ int xor_lane_i32(int v) {
  static_assert(S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32, "a power of two below 64");
  if constexpr (S == 1) return dpp_mov_i32<0xB1, 0xF>(v, v);  // quad_perm:[1,0,3,2]
  if constexpr (S == 2) return dpp_mov_i32<0x4E, 0xF>(v, v);  // quad_perm:[2,3,0,1]
  if constexpr (S == 4) return dpp_mov_i32<0x114, 0xA>(dpp_mov_i32<0x104, 0x5>(v, v), v);  // lanes 0-3 / 8-11 read +4, 4-7 / 12-15 read -4
  if constexpr (S == 8) return dpp_mov_i32<0x128, 0xF>(v, v);  // row_ror:8
  // Across rows: gfx950's row swaps.  v_permlane16_swap(a, b) exchanges a's odd rows with b's even rows, so with
  // a = b = v the first result is (r0, r0, r2, r2) and the second (r1, r1, r3, r3): a lane in an even row takes the
  // second, in an odd row the first.  v_permlane32_swap exchanges a's rows 2, 3 with b's rows 0, 1 likewise.
  if constexpr (S == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 16) ? r[0] : r[1]);
  }
  if constexpr (S == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32) ? r[0] : r[1]);
  }
  return v;
}
template <int S>
__device__ __forceinline__ double xor_lane(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)xor_lane_i32<S>((int)(unsigned)b), hi = (unsigned)xor_lane_i32<S>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum(double v) {  // the butterfly of `for (s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s)`
  v += xor_lane<32>(v);
  v += xor_lane<16>(v);
  v += xor_lane<8>(v);
  v += xor_lane<4>(v);
  v += xor_lane<2>(v);
  v += xor_lane<1>(v);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, xor_lane<32>(v));
  v = fmax(v, xor_lane<16>(v));
  v = fmax(v, xor_lane<8>(v));
  v = fmax(v, xor_lane<4>(v));
  v = fmax(v, xor_lane<2>(v));
  v = fmax(v, xor_lane<1>(v));
  return v;
}

// block reductions: wave shuffles, then one LDS exchange across the 4 waves
template <int NT = kThreads>
__device__ __forceinline__ double block_reduce_max(double v, double* red) {
  v = wave_max(v);
  const int t = threadIdx.x;
  __syncthreads();  // red may still be read by a previous user
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < NT / 64; ++i) r = fmax(r, red[i]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ int block_reduce_sum_int(int v, int* red) {
  v += xor_lane_i32<32>(v);
  v += xor_lane_i32<16>(v);
  v += xor_lane_i32<8>(v);
  v += xor_lane_i32<4>(v);
  v += xor_lane_i32<2>(v);
  v += xor_lane_i32<1>(v);
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  int r = 0;
  for (int i = 0; i < kThreads / 64; ++i) r += red[i];
  __syncthreads();
  return r;
}

// ---- whole-workgroup Jacobi: one barrier per round ---------------------------
// The eigenproblem sits on the critical path of every tree node, and a round of
// the textbook formulation costs ~3000 cycles here: a dependent div/sqrt/div/
// sqrt/div chain for (c, s), then three barrier-separated phases (rotation
// parameters, columns, rows).  This form removes all three costs:
//   * position-based tournament: the pair of "table" k always sits at matrix
//     positions (2k, 2k+1); after a round the rotated rows/columns are WRITTEN to
//     the positions the circle method moves them to, into a second buffer, so
//     every index is loop-invariant and read/write hazards need ONE barrier;
//   * every thread owns one 2x2 block (row pair kr, column pair kc) of A and of V
//     and derives the two rotations it needs itself from the three entries of
//     each pair (redundant, but no "few compute, all wait" phase);
//   * (c, s) from two reciprocal square roots (v_rsq_f64 + Newton) and no
//     division:  d = aqq-app, b = 2apq, h = hypot(d, b), cos(2t) = |d|/h,
//     c = sqrt((1+cos 2t)/2), s = sign(d) b / (2 h c)   (|t| <= pi/4, the same
//     rotation as the classical tau/t formula).
typedef __attribute__((address_space(3))) double lds_double;
using dh_eig::rsqrt_nr;
using dh_eig::jacobi_rotation;
using dh_eig::jacobi_dest;

// eigh of the symmetric D x D matrix in L.A: on return the diagonal of L.A holds
// the (unsorted) eigenvalues and the columns of L.V the eigenvectors.  Returns
// false if the input is not finite.  All kThreads threads.
__device__ bool jacobi_block(const Lds& L, int D) {
  const int t = threadIdx.x;
  const int LD = L.LD, JLD = L.JLD;
  if (L.j_alias) L.c_pts = nullptr;  // the work buffers overlay the point tile
  if (D == 1) {
    bool bad1 = false;
    if (t == 0) {
      L.V[0] = 1.0;
      bad1 = !isfinite(L.A[0]);
    }
    return !__syncthreads_or(bad1 ? 1 : 0);
  }
  const int m = (D + 1) / 2, P = 2 * m;
  // explicit LDS address space: through the struct the pointers are generic and the
  // round loop would be compiled to FLAT loads (several times the ds_read latency)
  lds_double* const ja0 = (lds_double*)L.JA[0];
  lds_double* const ja1 = (lds_double*)L.JA[1];
  lds_double* const jv0 = (lds_double*)L.JV[0];
  lds_double* const jv1 = (lds_double*)L.JV[1];
  // power-of-two pre-scaling to max|a_ij| in [1/2, 1): exact, undone on the eigenvalues
  bool bad = false;
  double amax = 0.0;
  for (int e = t; e < D * D; e += kThreads) {
    const int i = e / D, j = e - i * D;
    const double v = L.A[i * LD + j];
    if (!isfinite(v)) bad = true;
    amax = fmax(amax, fabs(v));
  }
  if (__syncthreads_or(bad ? 1 : 0)) return false;
  amax = block_reduce_max(amax, L.red);
  const int ex = amax > 0.0 ? ilogb(amax) + 1 : 0;
  for (int e = t; e < P * P; e += kThreads) {
    const int i = e / P, j = e - i * P;
    ja0[i * JLD + j] = (i < D && j < D) ? ldexp(L.A[i * LD + j], -ex) : 0.0;
    jv0[i * JLD + j] = (i == j && i < D) ? 1.0 : 0.0;
  }
  __syncthreads();
  // block ownership: slot 0 = block t, slot 1 = block t + kThreads (m <= 22 -> m^2 <= 2 kThreads).
  // EVERY lane derives the rotation of column pair (block index mod m) -- active block or
  // not -- so that the 64 consecutive block indices of a wave cover all pairs and the row
  // rotation can be fetched from a lane of the same wave (no LDS round trip, no barrier).
  const int nslots = m * m > kThreads ? 2 : 1;
  int ra[2], ca[2], dra[2], drb[2], dca[2], dcb[2], srcl[2];
  bool on[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int blk = t + sl * kThreads;
    on[sl] = blk < m * m;
    const int kr = blk / m, kc = blk - kr * m;
    const int wave_base = blk & ~63;
    int l = (on[sl] ? kr : 0) - wave_base % m;  // lane of this wave whose column pair is kr
    if (l < 0) l += m;
    srcl[sl] = l;
    ra[sl] = on[sl] ? 2 * kr : 0;
    ca[sl] = 2 * kc;
    dra[sl] = jacobi_dest(ra[sl], m);
    drb[sl] = jacobi_dest(ra[sl] + 1, m);
    dca[sl] = jacobi_dest(2 * kc, m);
    dcb[sl] = jacobi_dest(2 * kc + 1, m);
  }
  int cur = 0;
  int dpos = (D & 1) ? D : -1;  // position of the padding index (odd D)
  for (int sweep = 0; sweep < 60; ++sweep) {
    const lds_double* Ac = cur ? ja1 : ja0;
    double off = 0.0, dia = 0.0;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
      if (on[sl]) {
        const int r0 = ra[sl], c0 = ca[sl];
        const double x00 = Ac[r0 * JLD + c0], x01 = Ac[r0 * JLD + c0 + 1];
        const double x10 = Ac[(r0 + 1) * JLD + c0], x11 = Ac[(r0 + 1) * JLD + c0 + 1];
        if (r0 == c0) {
          dia = fma(x00, x00, fma(x11, x11, dia));
          off = fma(x01, x01, fma(x10, x10, off));
        } else {
          off = fma(x00, x00, fma(x01, x01, fma(x10, x10, fma(x11, x11, off))));
        }
      }
    off = wave_sum(off);
    dia = wave_sum(dia);
    if ((t & 63) == 0) {
      L.red[t >> 6] = off;
      L.red[8 + (t >> 6)] = dia;
    }
    __syncthreads();
    off = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    dia = L.red[8] + L.red[9] + L.red[10] + L.red[11];
    __syncthreads();
    if (!(off > 1e-31 * dia)) break;
    for (int r = 0; r < P - 1; ++r) {
      const lds_double* As = cur ? ja1 : ja0;
      const lds_double* Vs = cur ? jv1 : jv0;
      lds_double* Ad = cur ? ja0 : ja1;
      lds_double* Vd = cur ? jv0 : jv1;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        if (sl >= nslots) break;
        const int c0 = ca[sl], c1 = c0 + 1;
        double cc, sc;
        jacobi_rotation(As[c0 * JLD + c0], As[c1 * JLD + c1], As[c0 * JLD + c1], cc, sc);
        const double cr = __shfl(cc, srcl[sl]), sr = __shfl(sc, srcl[sl]);
        if (on[sl]) {
          const int r0 = ra[sl], r1 = r0 + 1;
          const double x00 = As[r0 * JLD + c0], x01 = As[r0 * JLD + c1];
          const double x10 = As[r1 * JLD + c0], x11 = As[r1 * JLD + c1];
          const double v00 = Vs[r0 * JLD + c0], v01 = Vs[r0 * JLD + c1];
          const double v10 = Vs[r1 * JLD + c0], v11 = Vs[r1 * JLD + c1];
          // rows: J_r^T x ; columns: (.) J_c      J = [[c, s], [-s, c]]
          const double y00 = cr * x00 - sr * x10, y01 = cr * x01 - sr * x11;
          const double y10 = sr * x00 + cr * x10, y11 = sr * x01 + cr * x11;
          double z00 = cc * y00 - sc * y01, z01 = sc * y00 + cc * y01;
          double z10 = cc * y10 - sc * y11, z11 = sc * y10 + cc * y11;
          if (r0 == c0) z01 = z10 = 0.0;  // the annihilated pair is exactly zero
          Ad[dra[sl] * JLD + dca[sl]] = z00;
          Ad[dra[sl] * JLD + dcb[sl]] = z01;
          Ad[drb[sl] * JLD + dca[sl]] = z10;
          Ad[drb[sl] * JLD + dcb[sl]] = z11;
          Vd[r0 * JLD + dca[sl]] = cc * v00 - sc * v01;
          Vd[r0 * JLD + dcb[sl]] = sc * v00 + cc * v01;
          Vd[r1 * JLD + dca[sl]] = cc * v10 - sc * v11;
          Vd[r1 * JLD + dcb[sl]] = sc * v10 + cc * v11;
        }
      }
      if (dpos >= 0) dpos = jacobi_dest(dpos, m);
      cur ^= 1;
      __syncthreads();
    }
  }
  // copy out: position a (skipping the padding) -> column a' of L.V, diagonal of L.A
  const lds_double* Af = cur ? ja1 : ja0;
  const lds_double* Vf = cur ? jv1 : jv0;
  for (int e = t; e < P * P; e += kThreads) {
    const int i = e / P, a = e - i * P;
    if (a == dpos || i >= D) continue;
    const int col = a - ((dpos >= 0 && a > dpos) ? 1 : 0);
    L.V[i * LD + col] = Vf[i * JLD + a];
    if (i == 0) L.A[col * LD + col] = ldexp(Af[a * JLD + a], ex);
  }
  __syncthreads();
  return true;
}

// ---- tile staging -----------------------------------------------------------
// tile[p][j] = f(pts[perm[start+p]][j]) for p < cnt ; f: 0 raw, 1 minus mean, 2 / scale
//
// The gather is latency-bound (one workgroup per CU, perm -> row -> LDS is a chain of two
// dependent global loads), so the loads are issued in batches of kStageBatch independent
// rows per thread before any of them is consumed.  The Lds struct remembers what the tile
// holds: a node that fits one tile is staged ONCE per kernel and reused by the mean / cov /
// fmax / k-means passes (raw -> centred is done in place).
// agent-scope (device-coherent) accesses: write-through stores, cache-bypassing loads
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent_i(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_i(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (round 6) A load under a condition -- `ok ? p[i] : 0.0` -- compiles to a branch around the load: exec mask saved
// (SGPRs spilled to VGPR lanes), the load alone in its block, `s_waitcnt 0` right behind it.  Every such load was a
// full LDS / L2 round trip of its own: the eight groups of the k-means label sums, 32 loads, took 6 000 cycles.  sel_ld
// reads UNCONDITIONALLY from an offset clamped to a valid one (base[0]) and selects afterwards, so the loads of a
// phase are issued together and waited for once.  Same values: same bits.
__device__ __forceinline__ double sel_ld(const double* base, int off, bool ok) {
  const double v = base[ok ? off : 0];
  return ok ? v : 0.0;
}
// through the L.coh switch
__device__ __forceinline__ double ld_c(const Lds& L, const double* p) { return L.coh ? ld_agent(p) : *p; }
__device__ __forceinline__ void st_c(const Lds& L, double* p, double v) {
  if (L.coh)
    st_agent(p, v);
  else
    *p = v;
}
__device__ __forceinline__ int ld_ci(const Lds& L, const int* p) { return L.coh ? ld_agent_i(p) : *p; }
__device__ __forceinline__ void st_ci(const Lds& L, int* p, int v) {
  if (L.coh)
    st_agent_i(p, v);
  else
    *p = v;
}

constexpr int kStageBatch = 8;
// SB rows of a thread are fetched together: their index loads, then their point loads, in flight at once -- two
// global round trips per SB * NT / DP points.  Measured in round 5 for k_ell's ellipsoid passes: SB = 16 / 32 (fewer
// round trips) spill 24 / 99 registers of a kernel that sits at 248, and IDX_LDS (the tile's indices fetched together
// and served from LDS: one round trip per batch) costs 15 spilled registers and a barrier per tile -- 1.325 ms per
// 64-run sequence against 1.296 without.  Both stay available, neither is used.
template <int NT = kThreads, int SB = kStageBatch, bool IDX_LDS = false>
__device__ __forceinline__ void stage_tile(const Lds& L, const double* __restrict__ pts,
                                           const int* __restrict__ perm, int start, int cnt, int D,
                                           int how) {
  // column j = t & (DP-1) (DP = pow2 >= D), rows stride by NT/DP: consecutive lanes
  // read consecutive doubles of one point (coalesced), no integer division.
  const int j = threadIdx.x & (L.DP - 1);
  const int p0 = threadIdx.x >> L.DPlog, pstep = NT >> L.DPlog;
  if (L.c_pts == pts && L.c_start == start && L.c_cnt == cnt) {
    if (L.c_how == how) return;  // the tile already holds exactly this (callers barrier after use)
    if (L.c_how == 0 && how == 1) {
      if (j < D) {
        const double mj = L.mean[j];
        for (int p = p0; p < cnt; p += pstep) L.tile[p * L.LD + j] -= mj;
      }
      L.c_how = 1;
      __syncthreads();
      return;
    }
  }
  // (round 6) A tile's permutation indices are fetched first, all together (one coalesced load or two a thread), and
  // served from LDS as 32-bit element offsets of the rows: a batch of rows then waits for ONE global round trip, its
  // points, instead of two (index, then row), a row's address is the scalar base plus a 32-bit lane offset, and with no
  // index registers to hold TEN rows of a thread are in flight instead of eight.  Every kernel of the pipeline starts
  // with a cold L2 (the XCDs' L2s are invalidated at a kernel boundary): a batch is a ~2 us trip to the memory side
  // whatever its instruction count, so what counts is rows in flight.  512-point tile: 16 trips -> 1 + 7.
  // Measured and dropped: `global_load_lds_dword` (a row = 2 D dwords straight into the tile, no registers, up to 63
  // rows in flight per wavefront): correct, but 540 cycles per row instruction -- 69 k cycles per 512-point tile
  // against 39 k; the 16-byte form that the matrix kernels use needs 16-byte aligned rows, a row here is 8 D bytes.
  // (one code path: a tile larger than the buffer goes in chunks of the buffer's size)
  const double mj = (how == 1 && j < D) ? L.mean[j] : 0.0;
  const double sj = (how == 2 && j < D) ? L.scale[j] : 1.0;
  constexpr int SB2 = 10;
  for (int c0 = 0; c0 < cnt; c0 += L.ibuf_cap) {
    const int cc = min(L.ibuf_cap, cnt - c0);
    if (c0) __syncthreads();  // (the previous chunk's indices have been read)
    for (int p = threadIdx.x; p < cc; p += NT) L.ibuf[p] = ld_ci(L, perm + start + c0 + p) * D;
    __syncthreads();
    if (j < D) {
      for (int pb = p0; pb < cc; pb += SB2 * pstep) {
        double x[SB2];
#pragma unroll
        for (int k = 0; k < SB2; ++k) {
          const int p = pb + k * pstep;
          x[k] = pts[(unsigned)(L.ibuf[p < cc ? p : 0] + j)];  // (clamped: unconditional loads, see sel_ld)
        }
#pragma unroll
        for (int k = 0; k < SB2; ++k) {
          const int p = pb + k * pstep;
          if (p < cc) {
            double v = x[k];
            if (how == 1) v -= mj;
            if (how == 2) v = v / sj;
            L.tile[(c0 + p) * L.LD + j] = v;
          }
        }
      }
    }
  }
  L.c_pts = pts;
  L.c_start = start;
  L.c_cnt = cnt;
  L.c_how = how;
  __syncthreads();
}

// mean of a node -> L.mean (np.mean(points, axis=0), bounding.py:1410).  Tile partials are reduced per
// tile and added in tile order: the grouping of the cooperative root (one part per tile, partials summed in
// part order), so that a live set gives the same bits whichever root routine its batch size selects.
__device__ __forceinline__ void node_mean(const Lds& L, const double* pts, const int* perm, int start, int count, int D) {
  const int t = threadIdx.x;
  const int G = kThreads / D > 0 ? kThreads / D : 1;  // point groups per dim
  const int j = t % D, g = t / D;
  double total = 0.0;
  for (int base = 0; base < count; base += L.TP) {
    const int cnt = min(L.TP, count - base);
    stage_tile(L, pts, perm, start + base, cnt, D, 0);
    double acc = 0.0;
    if (g < G && t < G * D)
      for (int p = g; p < cnt; p += G) acc += L.tile[p * L.LD + j];
    L.red[t] = acc;
    __syncthreads();
    if (t < D) {
      double s = 0.0;
      for (int gg = 0; gg < G; ++gg) s += L.red[gg * D + t];
      total += s;
    }
    __syncthreads();
  }
  if (t < D) L.mean[t] = total / (double)count;
  __syncthreads();
}

// population std of a node -> L.scale (points.std(axis=0), bounding.py:1503-1504); same tile-ordered
// grouping as node_mean
__device__ void node_std(const Lds& L, const double* pts, const int* perm, int start, int count, int D) {
  node_mean(L, pts, perm, start, count, D);
  const int t = threadIdx.x;
  const int G = kThreads / D > 0 ? kThreads / D : 1;
  const int j = t % D, g = t / D;
  double total = 0.0;
  for (int base = 0; base < count; base += L.TP) {
    const int cnt = min(L.TP, count - base);
    stage_tile(L, pts, perm, start + base, cnt, D, 1);
    double acc = 0.0;
    if (g < G && t < G * D)
      for (int p = g; p < cnt; p += G) {
        const double x = L.tile[p * L.LD + j];
        acc = fma(x, x, acc);
      }
    L.red[t] = acc;
    __syncthreads();
    if (t < D) {
      double s = 0.0;
      for (int gg = 0; gg < G; ++gg) s += L.red[gg * D + t];
      total += s;
    }
    __syncthreads();
  }
  if (t < D) L.scale[t] = sqrt(total / (double)count);
  __syncthreads();
}

// ---- the two GEMM-shaped passes run on the fp64 matrix cores -------------------
// v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]
// (one double each); the 16x16 result sits 4 per lane at row = (l>>4) + 4*reg, col = l&15.
// The point is not the FLOP rate (VALU fp64 is as fast) but operand traffic: the VALU forms
// of these loops read two LDS operands per FMA (~1250 ds_reads per point for the quadratic
// form), the MFMA forms read one operand pair per 1024 FMAs.
typedef double mfma_acc __attribute__((ext_vector_type(4)));
constexpr int kBarStride = 16;  // ints between part-barrier counters (one per 64-byte line)
constexpr int kMfmaMinDim = 10;  // below this the quadratic form stays on the VALU
#define DH_MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define DH_MFMA_F64_4X4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// sample covariance (ddof=1) of a node about L.mean -> L.A (np.cov, bounding.py:1411):
// C = Xc^T Xc.  Each wave contracts its own 64 points of every tile (K = points, 16 MFMA
// steps of 4), upper 16x16 blocks only; the four partial sums are folded in wave order.
// accumulate Xc^T Xc of the staged (centred) tile: each wave contracts its own 64 points
__device__ __forceinline__ void tile_cov_accumulate(const Lds& L, int cnt, int D, mfma_acc (&acc)[6], int wsel = -1) {
  const int t = threadIdx.x, lane = t & 63, w = wsel >= 0 ? wsel : (t >> 6), LD = L.LD;  // wsel: k_ell_wave plays the waves in turn
  const int nb = (D + 15) >> 4;  // 16-wide dimension blocks: 1..3 (D <= 44)
  const int lj = lane & 15, lk = lane >> 4;
  const bool v0 = lj < D, v1 = 16 + lj < D, v2 = 32 + lj < D;
  const int pend = min(cnt, w * 64 + 64);
  for (int p0 = w * 64; p0 < pend; p0 += 4) {
    const int p = p0 + lk;
    const bool pv = p < cnt;
    const int ro = p * LD + lj;
    const double f0 = sel_ld(L.tile, ro, pv && v0);
    acc[0] = DH_MFMA_F64(f0, f0, acc[0]);
    if (nb > 1) {
      const double f1 = sel_ld(L.tile, ro + 16, pv && v1);
      acc[1] = DH_MFMA_F64(f0, f1, acc[1]);
      acc[2] = DH_MFMA_F64(f1, f1, acc[2]);
      if (nb > 2) {
        const double f2 = sel_ld(L.tile, ro + 32, pv && v2);
        acc[3] = DH_MFMA_F64(f0, f2, acc[3]);
        acc[4] = DH_MFMA_F64(f1, f2, acc[4]);
        acc[5] = DH_MFMA_F64(f2, f2, acc[5]);
      }
    }
  }
}

// fold the four wave partials into the upper triangle of L.A in wave order (deterministic)
__device__ __forceinline__ void cov_fold_waves(const Lds& L, int D, const mfma_acc (&acc)[6]) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, LD = L.LD;
  const int nb = (D + 15) >> 4;
  const int lj = lane & 15, lk = lane >> 4;
  for (int wv = 0; wv < kThreads / 64; ++wv) {
    if (w == wv) {
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int ib = b == 0 ? 0 : b == 1 ? 0 : b == 2 ? 1 : b == 3 ? 0 : b == 4 ? 1 : 2;
        const int jb = b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 1 : 2;
        if (jb < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ib * 16 + lk + 4 * r, j = jb * 16 + lj;
            if (i < D && j < D) {
              double v = acc[b][r];
              if (wv > 0) v += L.A[i * LD + j];
              L.A[i * LD + j] = v;
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

// upper triangle of L.A times inv, mirrored
template <int NT = kThreads>
__device__ __forceinline__ void cov_finalize(const Lds& L, int D, double inv) {
  for (int e = threadIdx.x; e < D * D; e += NT) {
    const int i = e / D, j = e - i * D;
    if (i <= j) {
      const double c = L.A[i * L.LD + j] * inv;
      L.A[i * L.LD + j] = c;
      L.A[j * L.LD + i] = c;
    }
  }
  __syncthreads();
}

// (round 6) cov_fold_waves + cov_finalize + the copies that followed them, in two barriers instead of seven: every wave
// files its partial in a matrix of its own (wave 0 L.A, 1 L.V, 2 L.AM, 3 L.AX: all free while a node's covariance is
// formed), then the thread of an upper-triangle entry adds the four in wave order -- ((p0 + p1) + p2) + p3, the very
// sums the wave-by-wave fold forms -- scales, and writes the entry and its mirror to ALL FOUR matrices and to the
// node's global working copy: spd_fast then finds the covariance wherever it wants it (sweep source L.A or L.AM, the
// copy it keeps in L.AX or L.V) without a pass or a global round trip of its own.  Returns false if an entry is not
// finite (uniform).  Needs the four matrices to be distinct (the 256-thread carve).
__device__ __forceinline__ bool cov_fold_finalize_all(const Lds& L, int D, const mfma_acc (&acc)[6], double inv,
                                                      double* cov_g) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, LD = L.LD;
  const int nb = (D + 15) >> 4;
  const int lj = lane & 15, lk = lane >> 4;
  // (L.A, L.V, L.AM, L.AX are consecutive D x LD blocks of the 256-thread carve.  Written as a select among the four
  // struct members the compiler indexed the struct -- and kept the whole carve-up on the stack for it: 224 B of scratch)
  double* mine = L.A + (size_t)w * D * LD;
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const int ib = b == 0 ? 0 : b == 1 ? 0 : b == 2 ? 1 : b == 3 ? 0 : b == 4 ? 1 : 2;
    const int jb = b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 1 : 2;
    if (jb < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib * 16 + lk + 4 * r, j = jb * 16 + lj;
        if (i <= j && j < D) mine[i * LD + j] = acc[b][r];
      }
    }
  }
  __syncthreads();
  bool bad = false;
  for (int e = t; e < D * D; e += kThreads) {
    const int i = e / D, j = e - i * D;
    if (i <= j) {
      const int o = i * LD + j, ot = j * LD + i;
      const double c = (((L.A[o] + L.V[o]) + L.AM[o]) + L.AX[o]) * inv;
      if (!isfinite(c)) bad = true;
      L.A[o] = c;
      L.A[ot] = c;
      L.V[o] = c;
      L.V[ot] = c;
      L.AM[o] = c;
      L.AM[ot] = c;
      L.AX[o] = c;
      L.AX[ot] = c;
      cov_g[o] = c;
      cov_g[ot] = c;
    }
  }
  return __syncthreads_or(bad ? 1 : 0) == 0;
}

// tile_order: fold the four wave partials per TILE and add the tiles in order -- the grouping of the
// cooperative root (one part per tile, parts summed in part order), used for the root so that a live set
// gives the same bits whichever root routine its batch size selects.  Tree nodes keep the cheaper form
// (accumulators across all tiles, one fold): they are always built by this routine.
// fused_cov_g != nullptr (a tree node of the 256-thread kernels): the fold takes the fused form above and leaves the
// covariance in L.A, L.V, L.AM, L.AX and fused_cov_g; *fused_state = 1 (done, finite) or 2 (done, not finite).
__device__ __forceinline__ void node_cov(const Lds& L, const double* pts, const int* perm, int start, int count, int D,
                                         bool tile_order = false, double* fused_cov_g = nullptr, int* fused_state = nullptr) {
  mfma_acc acc[6];  // (0,0) (0,1) (1,1) (0,2) (1,2) (2,2)
#pragma unroll
  for (int b = 0; b < 6; ++b) acc[b] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
  // The staged tile may hold more than one 256-point tile (k_ell's top levels stage 512 points at once, round 5):
  // the arithmetic runs over the 256-point tiles inside it, in order -- the sums do not depend on the staging size
  const bool multi = tile_order && count > kThreads;
  for (int base = 0; base < count; base += L.TP) {
    const int cnt = min(L.TP, count - base);
    stage_tile(L, pts, perm, start + base, cnt, D, 1);
    for (int sub = 0; sub < cnt; sub += kThreads) {
      Lds LS = L;
      LS.tile = L.tile + (size_t)sub * L.LD;
      tile_cov_accumulate(LS, min(kThreads, cnt - sub), D, acc);
      if (multi) {
        __syncthreads();
        cov_fold_waves(L, D, acc);  // -> upper triangle of L.A
        for (int e = threadIdx.x; e < D * D; e += kThreads) {
          const int i = e / D, j = e - i * D;
          if (i <= j) L.V[i * L.LD + j] = ((base + sub) ? L.V[i * L.LD + j] : 0.0) + L.A[i * L.LD + j];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[b] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
      }
    }
    __syncthreads();
  }
  if (multi) {
    for (int e = threadIdx.x; e < D * D; e += kThreads) {
      const int i = e / D, j = e - i * D;
      if (i <= j) L.A[i * L.LD + j] = L.V[i * L.LD + j];
    }
    __syncthreads();
  } else if (fused_cov_g != nullptr && L.V == L.A + D * L.LD && L.AM == L.V + D * L.LD && L.AX == L.AM + D * L.LD) {
    *fused_state = cov_fold_finalize_all(L, D, acc, 1.0 / (double)(count - 1), fused_cov_g) ? 1 : 2;
    return;
  } else {
    cov_fold_waves(L, D, acc);
  }
  cov_finalize(L, D, 1.0 / (double)(count - 1));
}

// max over the staged tile of x^T AM x (AM: D x LD in LDS, symmetric; x = tile rows):
// Z = X AM on the matrix cores (M = 16 points per block, N = dimension blocks, K = D in
// steps of 4), then the row-wise dot Z.x and a 16-lane reduction.  Returns the per-thread
// running maximum (callers reduce over the block).
//
// Round 5: the B operand (AM's fragments) is the same for every block of points, so a lane holds its fragments in
// registers for the whole node (QuadB: up to 11 K steps x 3 dimension blocks) instead of re-reading them from LDS at
// every step; a block's A operands (its K steps of x) are fetched together, one LDS round trip; and a wavefront keeps
// TWO blocks of 16 points in flight -- four to six independent accumulation chains on the matrix pipe where one
// block's two were each waiting for their own previous step.  Per accumulator the K steps still run in ascending
// order: the same bits.
// Register shape: NB dimension blocks of 16, at most KS K steps of 4.  Built for <2, 8> (17 <= D <= 32): the other
// shapes keep the general form below -- three copies of this loop at k_ell's every call site cost more in spills
// than they gave.
#ifndef DH_QUAD_TWO
#define DH_QUAD_TWO false
#endif
template <int NB, int KS>
struct QuadB {
  double b[KS][NB];
};
template <int NB, int KS>
__device__ __forceinline__ void quad_load_b(const double* AM, int D, int LD, QuadB<NB, KS>& B) {
  const int lane = threadIdx.x & 63, lj = lane & 15, lk = lane >> 4;
  const int ksteps = (D + 3) >> 2;
  const int bo = lk * LD + lj;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const bool kv = ks < ksteps && ks * 4 + lk < D;
#pragma unroll
    for (int n = 0; n < NB; ++n) B.b[ks][n] = sel_ld(AM, bo + ks * 4 * LD + 16 * n, kv && 16 * n + lj < D);
  }
}
// one block of 16 points after its products: row-wise dot with x, 16-lane sums, running maximum
template <int NB>
__device__ __forceinline__ double quad_block_max(const double* tile, int LD, int p0, int cnt, int D, const mfma_acc (&z)[NB], double best) {
  const int lane = threadIdx.x & 63, lj = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = p0 + lk + 4 * r;
    const bool pv = p < cnt;
    const int xo = p * LD + lj;
    const bool c0 = pv && lj < D;
    const double x0 = sel_ld(tile, xo, c0);
    double sacc = c0 ? z[0][r] * x0 : 0.0;
    if constexpr (NB > 1) {
      const bool c1 = pv && 16 + lj < D;
      const double x1 = sel_ld(tile, xo + 16, c1);
      sacc = c1 ? fma(z[1][r], x1, sacc) : sacc;
    }
    if constexpr (NB > 2) {
      const bool c2 = pv && 32 + lj < D;
      const double x2 = sel_ld(tile, xo + 32, c2);
      sacc = c2 ? fma(z[2][r], x2, sacc) : sacc;
    }
    sacc += xor_lane<1>(sacc);
    sacc += xor_lane<2>(sacc);
    sacc += xor_lane<4>(sacc);
    sacc += xor_lane<8>(sacc);
    if (pv) best = fmax(best, sacc);
  }
  return best;
}
template <int NT, int NB, int KS>
__device__ __forceinline__ double tile_quadform_max(const double* tile, int LD, const QuadB<NB, KS>& B, int cnt, int D, double best) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ksteps = (D + 3) >> 2;
  const int lj = lane & 15, lk = lane >> 4;
  constexpr int NW = NT / 64;
  constexpr bool TWO = DH_QUAD_TWO;
  for (int mb = w; mb * 16 < cnt; mb += (TWO ? 2 : 1) * NW) {
    const int pA = mb * 16, pB = (mb + NW) * 16;
    const bool hasB = TWO && pB < cnt;  // (uniform per wavefront)
    const bool va = pA + lj < cnt, vb = hasB && pB + lj < cnt;
    const int xao = (pA + lj) * LD + lk, xbo = (pB + lj) * LD + lk;
    double aA[KS], aB[TWO ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool kv = ks < ksteps && ks * 4 + lk < D;
      aA[ks] = sel_ld(tile, xao + ks * 4, va && kv);
      if constexpr (TWO) aB[ks] = sel_ld(tile, xbo + ks * 4, vb && kv);
    }
    mfma_acc zA[NB], zB[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) zA[n] = zB[n] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks < ksteps) {
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          zA[n] = DH_MFMA_F64(aA[ks], B.b[ks][n], zA[n]);
          if constexpr (TWO)
            if (hasB) zB[n] = DH_MFMA_F64(aB[ks], B.b[ks][n], zB[n]);
        }
      }
    }
    best = quad_block_max<NB>(tile, LD, pA, cnt, D, zA, best);
    if constexpr (TWO)
      if (hasB) best = quad_block_max<NB>(tile, LD, pB, cnt, D, zB, best);
  }
  return best;
}
// the general form (any D >= kMfmaMinDim): both operands from LDS at every K step, one block of points at a time
template <int NT = kThreads>
__device__ __forceinline__ double tile_quadform_max_lds(const Lds& L, const double* AM, int cnt, int D, double best) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, LD = L.LD;
  const int nb = (D + 15) >> 4, ksteps = (D + 3) >> 2;
  const int lj = lane & 15, lk = lane >> 4;
  for (int mb = w; mb * 16 < cnt; mb += NT / 64) {
    const int p0 = mb * 16;
    mfma_acc z0 = {0.0, 0.0, 0.0, 0.0}, z1 = z0, z2 = z0;
    const bool pa = p0 + lj < cnt;
    const int xro = (p0 + lj) * LD + lk, bro = lk * LD + lj;
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = ks * 4 + lk;
      const bool kv = k < D;
      const double a = sel_ld(L.tile, xro + ks * 4, pa && kv);
      const double b0 = sel_ld(AM, bro + ks * 4 * LD, kv && lj < D);
      z0 = DH_MFMA_F64(a, b0, z0);
      if (nb > 1) {
        const double b1 = sel_ld(AM, bro + ks * 4 * LD + 16, kv && 16 + lj < D);
        z1 = DH_MFMA_F64(a, b1, z1);
        if (nb > 2) {
          const double b2 = sel_ld(AM, bro + ks * 4 * LD + 32, kv && 32 + lj < D);
          z2 = DH_MFMA_F64(a, b2, z2);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + lk + 4 * r;
      const bool pv = p < cnt;
      const int xo = p * LD + lj;
      const bool c0 = pv && lj < D, c1 = nb > 1 && pv && 16 + lj < D, c2 = nb > 2 && pv && 32 + lj < D;
      const double x0 = sel_ld(L.tile, xo, c0), x1 = sel_ld(L.tile, xo + 16, c1), x2 = sel_ld(L.tile, xo + 32, c2);
      double sacc = c0 ? z0[r] * x0 : 0.0;
      sacc = c1 ? fma(z1[r], x1, sacc) : sacc;
      sacc = c2 ? fma(z2[r], x2, sacc) : sacc;
      sacc += xor_lane<1>(sacc);
      sacc += xor_lane<2>(sacc);
      sacc += xor_lane<4>(sacc);
      sacc += xor_lane<8>(sacc);
      if (pv) best = fmax(best, sacc);
    }
  }
  return best;
}
// the quadratic-form maximum of the staged tile (D >= kMfmaMinDim): the register form where it is built (256-thread
// workgroups, 17 <= D <= 32), the general form otherwise -- per point the same operations in the same order
template <int NT = kThreads>
__device__ __forceinline__ double tile_quadform_max(const Lds& L, const double* AM, int cnt, int D, double best) {
  if (NT == kThreads && D > 16 && D <= 32) {
    QuadB<2, 8> B;
    quad_load_b<2, 8>(AM, D, L.LD, B);
    return tile_quadform_max<NT, 2, 8>(L.tile, L.LD, B, cnt, D, best);
  }
  return tile_quadform_max_lds<NT>(L, AM, cnt, D, best);
}

// max_i delta_i^T AM delta_i over a node (bounding.py:1438), delta about L.mean.  Tiles are taken last to first: the
// covariance pass before left its last tile staged (a maximum does not care about the order)
template <int NT = kThreads>
__device__ __forceinline__ double node_fmax(const Lds& L, const double* pts, const int* perm, int start, int count, int D) {
  double best = -INFINITY;
  PH_T0();
  for (int base = ((count - 1) / L.TP) * L.TP; base >= 0; base -= L.TP) {
    const int cnt = min(L.TP, count - base);
    stage_tile<NT>(L, pts, perm, start + base, cnt, D, 1);
    PH_ADD(5);
    if (D >= kMfmaMinDim) {
      best = tile_quadform_max<NT>(L, L.AM, cnt, D, best);  // (any number of staged points: a maximum has no order)
    } else {
      // small D: a point's D^2 FMAs are cheaper than the 16-lane reductions of the MFMA form
      for (int p = threadIdx.x; p < cnt; p += NT) {
        const double* x = L.tile + p * L.LD;
        double q = 0.0;
        for (int i = 0; i < D; ++i) {
          double r = 0.0;
          const double* row = L.AM + i * L.LD;
          for (int j = 0; j < D; ++j) r = fma(row[j], x[j], r);
          q = fma(x[i], r, q);
        }
        best = fmax(best, q);
      }
    }
    __syncthreads();
  }
  return block_reduce_max<NT>(best, L.red);
}

// helpers on D x D LDS matrices (all threads)
__device__ __forceinline__ void mat_from_eig(const Lds& L, double* out, const double* lamv, int D,
                                             bool inverse) {
  // out = (V * w) V^T with w = lam or 1/lam   ((eigvec * x) @ eigvec.T)
  if (threadIdx.x < D) L.red[threadIdx.x] = inverse ? 1.0 / lamv[threadIdx.x] : lamv[threadIdx.x];
  __syncthreads();
  for (int e = threadIdx.x; e < D * D; e += kThreads) {
    const int i = e / D, j = e % D;
    double s = 0.0;
    for (int k = 0; k < D; ++k) s = fma(L.V[i * L.LD + k] * L.red[k], L.V[j * L.LD + k], s);
    out[i * L.LD + j] = s;
  }
  __syncthreads();
}

// improve_covar_mat (bounding.py:1311-1384) on L.A (the covariance, kept in
// L.AX-independent copy `cov`), producing L.AM (precision), L.AX (axes), L.lam
// (eigenvalues of the returned covariance).  `cov` is a D x LD global/LDS buffer
// holding the matrix to regularise; it is updated in place.  Returns good_mat.
// WAVE = true: the eigen-solves run on wave 0 with the compact wave-level Jacobi of eig_wave.h (three LDS
// passes per round: several times slower than jacobi_block, but a fraction of its registers).  Used where
// the route is a rare fallback inside a kernel that must stay small (k_ell<false>).
template <bool WAVE = false>
__device__ __forceinline__ bool regularize(const Lds& L, double* cov, int D) {
  const int t = threadIdx.x;
  int failed = 0;
  int trial = 0;
  PH_T0();
  for (trial = 0; trial < kNTries; ++trial) {
    failed = 0;
    // eigh(cov): copy into A, solve with wave 0
    for (int e = t; e < D * D; e += kThreads) L.A[(e / D) * L.LD + e % D] = cov[(e / D) * L.LD + e % D];
    __syncthreads();
    PH_ADD(8);
    bool fin;
    if constexpr (WAVE) {
      int ok = 1;
      if (t < 64) ok = jacobi_wave(L.A, L.V, D, L.LD, L.rc, L.rs, L.ri) ? 1 : 0;
      fin = __syncthreads_and(ok) != 0;
    } else {
      // (jacobi_block is a call: handed a COPY of the carve-up, the original stays in registers -- with its address
      // taken every L.xxx of the calling kernel was a scratch load and every LDS access through it a FLAT one:
      // k_root_parts compiled to 1 132 flat loads and 448 B of scratch a lane, round 6)
      const Lds Lc = L;
      fin = jacobi_block(Lc, D);
      if (L.j_alias) L.c_pts = nullptr;  // (what the callee noted on its copy: the work buffers overlay the point tile)
    }
    PH_ADD(6);
    if (fin && t < 64) sort_eigs_wave(L.A, L.V, L.lam, L.perm_sort, L.AX, D, L.LD);
    __syncthreads();
    PH_ADD(7);
    double top = -INFINITY, bot = INFINITY;
    bool allfin = fin;
    if (fin) {
      for (int k = 0; k < D; ++k) {
        const double l = L.lam[k];
        if (!isfinite(l)) allfin = false;
        top = fmax(top, l);
        bot = fmin(bot, l);
      }
    }
    if (allfin) {
      if (top <= 0.0)
        failed = 2;
      else if (bot < top / kMaxCond)
        failed = 1;
      else {
        // axes = eigvec * eigval**.5
        for (int e = t; e < D * D; e += kThreads) {
          const int i = e / D, k = e % D;
          L.AX[i * L.LD + k] = L.V[i * L.LD + k] * sqrt(L.lam[k]);
        }
        __syncthreads();
        break;
      }
    } else {
      failed = 2;
    }
    if (failed == 1) {
      // eigval_fix = max(eigval, eig_mult * maxval / max_condition_number)
      __syncthreads();
      if (t < D) L.lam[t] = fmax(L.lam[t], kEigMult * top / kMaxCond);
      __syncthreads();
      mat_from_eig(L, cov, L.lam, D, false);
    } else {
      const double coeffmin = 1e-10;
      const double coeff = coeffmin * pow(1.0 / coeffmin, (double)trial / (double)(kNTries - 1));
      for (int e = t; e < D * D; e += kThreads) {
        const int i = e / D, j = e % D;
        cov[i * L.LD + j] = (1.0 - coeff) * cov[i * L.LD + j] + coeff * (i == j ? 1.0 : 0.0);
      }
      __syncthreads();
    }
  }
  if (failed > 0) {
    // identity fallback (bounding.py:1374-1378)
    for (int e = t; e < D * D; e += kThreads) {
      const int i = e / D, j = e % D;
      const double v = (i == j) ? 1.0 : 0.0;
      cov[i * L.LD + j] = v;
      L.AM[i * L.LD + j] = v;
      L.AX[i * L.LD + j] = v;
    }
    if (t < D) L.lam[t] = 1.0;
    __syncthreads();
    return false;  // trial == ntries-1 != 0
  }
  PH_ADD(8);
  mat_from_eig(L, L.AM, L.lam, D, true);
  PH_ADD(9);
  return trial == 0;
}

// ---- the eigen-free path of bounding_ellipsoid for tree nodes ---------------------------------
// MultiEllipsoid.update builds ~50 tree nodes per live set and keeps one to a dozen of them.  What a
// node needs on the way is (bounding.py:1387-1461, 1464-1563): the verdict of improve_covar_mat
// ("is the condition number below 1e12?"), the precision matrix for the Mahalanobis maximum, the
// log-volume for the accept test and -- only if it is big enough to be split -- its major axis for
// the k-means seeds.  None of that requires the full eigen-decomposition, which was 60 % of the
// rebuild (a 25 x 25 Jacobi solve is ~175 dependent rounds = 0.1 ms on every level of the tree):
//   * LDL^T (Cholesky without roots) of the covariance: positive pivots <=> positive definite,
//     ln det = sum ln pivots, cov^-1 = M^T D^-1 M with M = L^-1 by forward substitution;
//   * cond(cov) <= tr(cov) tr(cov^-1): if that bound is below kFastCond the reference's test
//     lam_min < lam_max / 1e12 is certainly false (the matrix is "good" and stays untouched), and
//     the explicit inverse is accurate to cond * eps <= 1e-9;
//   * the dominant eigenvector by repeated squaring B <- B^2 / scale on the matrix cores: after k
//     squarings the weight of the second eigenvalue is (lam_2 / lam_1)^(2^k); tr(B^2) / tr(B)^2 = sum of
//     the squared weights says when it has vanished.  lam_max is the Rayleigh quotient.
// Anything else -- a pivot <= 0, a bound above kFastCond, a pair of leading eigenvalues closer than
// ~1e-11 relative (no convergence in kFastSquarings) -- returns false and the caller takes the
// reference's own route (regularize: Jacobi eigh + the 100-trial loop).  The ellipsoids that
// survive the accept test get their full eigen-system from k_out_eig at the end (same Jacobi, run
// once per OUTPUT instead of once per tree node).
#ifndef DH_SPD_OVERLAP
// 1: the sweeps (wavefronts 0, 1) and the squarings (wavefronts 2, 3) of spd_fast side by side.  MEASURED AND OFF
// (round 6, tools/r6_phase.py): bit-identical, but the sweeps are bound by the instructions of a wavefront, not by
// latency -- on 128 threads a double sweep takes twice as long (ldl+inv 270 -> 520-700 per node, in hundreds of cycles)
// and eats what the hidden squarings give (-25 .. -330): 64 runs 1.04 -> 1.09 ms.  The code stays for the record.
#define DH_SPD_OVERLAP 0
#endif
constexpr double kFastCond = 1e7;
constexpr int kFastSquarings = 48;

// trace of a D x D LDS matrix, computed redundantly by every wave (D <= 44 < 64)
__device__ __forceinline__ double wave_trace(const double* M, int D, int LD) {
  const int lane = threadIdx.x & 63;
  return wave_sum(sel_ld(M, lane * LD + lane, lane < D));
}

// Q = s2 * P P for the symmetric D x D matrix P (LDS, D x LD); all threads; caller barriers
template <int NT = kThreads>
__device__ __forceinline__ void sym_square(const double* P, double* Q, int D, int LD, double s2) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (D >= kMfmaMinDim) {
    const int nb = (D + 15) >> 4, ksteps = (D + 3) >> 2;
    const int lj = lane & 15, lk = lane >> 4;
    for (int tile = w; tile < nb * nb; tile += NT / 64) {
      const int ti = tile / nb, tj = tile - ti * nb;
      mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
      const int ca = ti * 16 + lj, cb = tj * 16 + lj;
      for (int ks = 0; ks < ksteps; ++ks) {
        const int k = ks * 4 + lk;
        const bool kv = k < D;
        // A[i][k] = P[k][i] (symmetric): both operands are row reads, lanes along the row
        const double av = sel_ld(P, k * LD + ca, kv && ca < D);
        const double bv = sel_ld(P, k * LD + cb, kv && cb < D);
        acc = DH_MFMA_F64(av, bv, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + lk + 4 * r;
        if (i < D && cb < D) Q[i * LD + cb] = acc[r] * s2;
      }
    }
  } else {
    for (int e = t; e < D * D; e += NT) {
      const int i = e / D, j = e - i * D;
      double sum = 0.0;
      for (int k = 0; k < D; ++k) sum = fma(P[k * LD + i], P[k * LD + j], sum);
      Q[i * LD + j] = sum * s2;
    }
  }
}

// sym_square's matrix-core form for the wavefronts wfirst .. wfirst + nw - 1 alone (tile by tile the same products:
// the same bits as sym_square's); D >= kMfmaMinDim; caller barriers
__device__ __forceinline__ void sym_square_waves(const double* P, double* Q, int D, int LD, double s2, int wfirst, int nw) {
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) - wfirst;
  const int nb = (D + 15) >> 4, ksteps = (D + 3) >> 2;
  const int lj = lane & 15, lk = lane >> 4;
  for (int tile = w; tile < nb * nb; tile += nw) {
    const int ti = tile / nb, tj = tile - ti * nb;
    mfma_acc acc = {0.0, 0.0, 0.0, 0.0};
    const int ca = ti * 16 + lj, cb = tj * 16 + lj;
    for (int ks = 0; ks < ksteps; ++ks) {
      const int k = ks * 4 + lk;
      const bool kv = k < D;
      const double av = sel_ld(P, k * LD + ca, kv && ca < D);
      const double bv = sel_ld(P, k * LD + cb, kv && cb < D);
      acc = DH_MFMA_F64(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ti * 16 + lk + 4 * r;
      if (i < D && cb < D) Q[i * LD + cb] = acc[r] * s2;
    }
  }
}

// 1 / x to full precision: v_rcp_f64 (2^-24, tools/micro/rsq_acc.hip) + two Newton steps
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) y = fma(y, fma(-x, y, 1.0), y);
  return y;
}

// in: cov (D x LD, global or LDS; untouched).  out (on true): L.AM = cov^-1, *logdet = ln det cov,
// and with want_axis L.AX[:, 0] = sqrt(lam_max) v_max (canonical sign), L.lam[0] = lam_max, every
// other column of L.AX and entry of L.lam zero.  Work space: L.A, L.V, L.AX, L.red.  All threads;
// the return value is uniform.
//
// The inverse is formed by D sweeps (the sweep operator of regression codes: Gauss-Jordan on a
// symmetric matrix without pivoting -- for a positive definite matrix the pivots are the positive
// Schur-complement pivots of LDL^T, so their logarithms sum to ln det).  Sweep k reads one buffer and
// writes the other (L.AM <-> L.A), so a sweep is: pivot, pivot row / column, own entries, ONE barrier.
// Thread map: column j = t mod JW, rows t / JW, t / JW + 256 / JW, ... (JW = 32 or 64 >= D).
// (round 6) The covariance is taken from LDS: L.A holds it on entry (every caller has just formed it there); `cov`,
// the node's global working copy, is only read where the carve has no room for a copy.  prepared = true: the caller's
// fused fold (cov_fold_finalize_all) has already put it into L.A, L.V, L.AM and L.AX and checked that it is finite.
// *cov_keep (on true): an LDS matrix that still holds the untouched covariance, or nullptr (then `cov` does).
template <int NT = kThreads>
__device__ __forceinline__ bool spd_fast(const Lds& L, const double* cov, int D, bool want_axis, double* logdet,
                                         const double** cov_keep = nullptr, bool prepared = false) {
  const int t = threadIdx.x, LD = L.LD;
  // Thread map of the element-wise passes (round 6): column j = t mod D, rows t / D, t / D + NT / D, ... -- every
  // thread of the first (NT / D) D has work and a thread's rows are ceil(D / (NT / D)): 3 at D = 25 and 256 threads
  // where the power-of-two map (column t & 31, 8 row groups) executed 4 row slots for 25 rows, the fourth for the
  // threads of one group only.  The sweeps are instruction-bound (one instruction per ~10 cycles and wavefront): the
  // slots executed are their time.  `j` is D for the threads beyond, which every `j < D` below already excludes.
  const int istep = NT / D > 0 ? NT / D : 1;
  const int i0 = t / D;
  const int j = i0 < istep ? t - i0 * D : D;
  const int jsh = D <= 32 ? 5 : 6;  // (the overlap experiment's 128-thread map)
  // Two sweeps per barrier (round 5): sweep k + 1 needs, of the matrix sweep k produces, row k + 1, column k + 1 and
  // the entry itself -- each of them one fma of entries of the matrix sweep k READS, so a thread forms them on its own
  // (the very expressions sweep k would have stored) and applies both sweeps to its entries before anyone has to
  // wait: ceil(D / 2) barriers and LDS round-trip chains instead of D, bit for bit the same inverse.
  const int nswap = (D + 1) >> 1;
  double* src = (nswap & 1) ? L.AM : L.A;  // nswap buffer changes later the result sits in L.A
  double* dst = (nswap & 1) ? L.A : L.AM;
  double* pivs = L.red;  // the D pivots
  // the copy of the covariance that outlives the sweeps (they work in L.A / L.AM): L.AX while the squarings need L.A and
  // L.V, else L.V; none where the carve has no separate matrices (k_ell_wave's leaves: no axis wanted either)
  double* keep = (L.V != nullptr && L.AX != L.A) ? (want_axis ? L.AX : L.V) : nullptr;
  if (cov_keep) *cov_keep = keep;
  PH_T0();
  if (!prepared) {
    bool bad = false;
    if (j < D)
      for (int i = i0; i < D; i += istep) {
        const double c = L.A[i * LD + j];
        if (!isfinite(c)) bad = true;
        if (src != L.A) src[i * LD + j] = c;
        if (keep) keep[i * LD + j] = c;
      }
    if (__syncthreads_or(bad ? 1 : 0)) return false;
  }
  // (the squarings and the Rayleigh quotient read the covariance from `keep` where there is one, else from `cov`: two
  // code paths, not one pointer -- a pointer that may be LDS or global is a generic one, and its loads FLAT loads)
  const double tr_cov = wave_trace(src, D, LD);
  // (round 6) SWEEPS AND SQUARINGS SIDE BY SIDE.  The inverse (13 double sweeps at D = 25) and the dominant eigenvector
  // (8-14 squarings) both need only the covariance, and they ran one after the other: 12 + 8-16 us of a node's ~45.
  // In the 256-thread kernels wavefronts 0, 1 now sweep (their 128 threads take the rows in two batches) while
  // wavefronts 2, 3 square -- first from the untouched copy in L.AX (scaled by s0^2 in the product: powers of two,
  // the same bits as squaring the scaled copy), then between L.V and L.AX -- meeting at the sweeps' own barriers.  A
  // squaring chain that is not done when the sweeps are goes on with all four wavefronts, as before.  The covariance
  // is read back from the node's global working copy afterwards (one trip) for the Rayleigh quotient and the record.
  const int wv = threadIdx.x >> 6;
  const bool ovl = NT == kThreads && want_axis && keep != nullptr && keep == L.AX && D <= 32 && D >= kMfmaMinDim &&
                   !(DH_SPD_OVERLAP == 0);
  const bool sweeper = !ovl || (wv < 2 && i0 < (128 / D > 0 ? 128 / D : 1));
  const int istep_e = ovl ? (128 / D > 0 ? 128 / D : 1) : istep;
  const int nslots = (D + istep_e - 1) / istep_e;  // row slots of a thread (uniform)
  (void)jsh;
  const double s0_o = ldexp(1.0, -(ilogb(tr_cov) + 1));  // (as below: trace of the scaled copy in [1/2, 1))
  const double* sP = L.AX;
  double* sQ = L.V;
  double trP_o = tr_cov * s0_o;
  bool conv_o = false;
  int nsq = 0;
  auto square_issue = [&]() {  // wavefronts 2, 3: one squaring's products
    if (ovl && wv >= 2 && !conv_o) {
      const double sc = ldexp(1.0, -(ilogb(trP_o * trP_o) + 1));
      sym_square_waves(sP, sQ, D, LD, nsq == 0 ? sc * s0_o * s0_o : sc, 2, 2);
    }
  };
  auto square_close = [&]() {  // ... and, behind the barrier, its trace and the convergence test
    if (ovl && wv >= 2 && !conv_o) {
      const double sc = ldexp(1.0, -(ilogb(trP_o * trP_o) + 1));
      const double trQ = wave_trace(sQ, D, LD);
      const double r = trQ / (trP_o * trP_o * sc);
      conv_o = 1.0 - r < 1e-8;
      trP_o = trQ;
      const double* nP = sQ;
      sQ = nsq == 0 ? L.AX : const_cast<double*>(sP);
      sP = nP;
      ++nsq;
    }
  };
  bool ok = true;
  int k = 0;
  for (; k + 1 < D; k += 2) {
    const int k1 = k + 1;
    const double piv = src[k * LD + k];
    const double a01 = src[k * LD + k1], a10 = src[k1 * LD + k], a11 = src[k1 * LD + k1];
    if (!(piv > 0.0) || !isfinite(piv)) {
      ok = false;  // uniform: every thread reads the same words
      break;
    }
    const double rp = rcp_nr(piv);
    const double c01r = a01 * rp;                  // sweep k: (row k, column k + 1)
    const double piv1 = fma(-a10, c01r, a11);      // sweep k: (k + 1, k + 1) = the next pivot
    if (!(piv1 > 0.0) || !isfinite(piv1)) {
      ok = false;
      break;
    }
    const double rp1 = rcp_nr(piv1);
    if (t == 0) {
      pivs[k] = piv;
      pivs[k1] = piv1;
    }
    if (sweeper && j < D) {
      const double cj = src[k * LD + j], dj = src[k1 * LD + j];
      const double cjr = cj * rp;
      const double r1j = (j == k) ? a10 * rp : fma(-a10, cjr, dj);  // sweep k: (row k + 1, column j)
      const double cjr1 = r1j * rp1;
      // (U row slots of a thread as straight-line code: exactly as many as the map needs -- see above)
      auto rows = [&](auto Uc, int ib) {
        constexpr int U = decltype(Uc)::value;
        double ci[U], ei[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = ib + u * istep_e;
          const int ic = i < D ? i : k;
          ci[u] = src[ic * LD + k];
          ei[u] = src[ic * LD + k1];
          w[u] = src[ic * LD + j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = ib + u * istep_e;
          // sweep k at (i, j) and at (i, k + 1)
          double v = fma(-ci[u], cjr, w[u]);
          v = (j == k) ? ci[u] * rp : v;
          v = (i == k) ? ((j == k) ? -rp : cjr) : v;
          const double e1 = (i == k) ? c01r : fma(-ci[u], c01r, ei[u]);
          // sweep k + 1 at (i, j)
          double val = fma(-e1, cjr1, v);
          val = (j == k1) ? e1 * rp1 : val;
          val = (i == k1) ? ((j == k1) ? -rp1 : cjr1) : val;
          if (i < D) dst[i * LD + j] = val;
        }
      };
      if (nslots == 1)
        rows(std::integral_constant<int, 1>{}, i0);
      else if (nslots == 2)
        rows(std::integral_constant<int, 2>{}, i0);
      else if (nslots == 3)
        rows(std::integral_constant<int, 3>{}, i0);
      else
        for (int ib = i0; ib < D; ib += 4 * istep_e) rows(std::integral_constant<int, 4>{}, ib);
    }
    square_issue();
    __syncthreads();
    square_close();
    double* tmp = src;
    src = dst;
    dst = tmp;
  }
  if (ok && k < D) {  // D odd: the last sweep alone
    const double piv = src[k * LD + k];
    if (!(piv > 0.0) || !isfinite(piv)) {
      ok = false;
    } else {
      const double rp = rcp_nr(piv);
      if (t == 0) pivs[k] = piv;
      if (sweeper && j < D) {
        const double cj = src[k * LD + j];
        const double cjr = cj * rp;
        for (int ib = i0; ib < D; ib += 4 * istep_e) {
          double ci[4], w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = ib + u * istep_e;
            const int ic = i < D ? i : k;
            ci[u] = src[ic * LD + k];
            w[u] = src[ic * LD + j];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = ib + u * istep_e;
            double val = fma(-ci[u], cjr, w[u]);
            val = (j == k) ? ci[u] * rp : val;
            val = (i == k) ? ((j == k) ? -rp : cjr) : val;
            if (i < D) dst[i * LD + j] = val;
          }
        }
      }
      square_issue();
      __syncthreads();
      square_close();
      double* tmp = src;
      src = dst;
      dst = tmp;
    }
  }
  if (!ok) return false;
  // the squaring wavefronts' state to everybody (read behind the barrier of the symmetrised inverse below)
  __shared__ double s_ovl_tr;
  __shared__ int s_ovl[3];
  if (ovl && threadIdx.x == 128) {
    s_ovl_tr = trP_o;
    s_ovl[0] = conv_o ? 1 : 0;
    s_ovl[1] = nsq;
    s_ovl[2] = sP == L.V ? 1 : 0;
  }
  double ld;
  {
    const int lane = t & 63;  // every wave for itself (D <= 44 < 64)
    ld = wave_sum(lane < D ? log(pivs[lane]) : 0.0);
  }
  // AM = -(result), symmetrised (the two triangles differ by rounding); result in L.A, AM elsewhere
  if (j < D)
    for (int i = i0; i < D; i += istep) L.AM[i * LD + j] = -0.5 * (L.A[i * LD + j] + L.A[j * LD + i]);
  __syncthreads();
  const double tr_am = wave_trace(L.AM, D, LD);
  // cond <= tr(cov) tr(cov^-1); NaN fails the comparison as well
  if (!(tr_cov * tr_am < kFastCond)) return false;
  *logdet = ld;
  PH_ADD(13);
  // (with the copy kept in L.AX the matrix is zeroed where the axis is written, at the end)
  if (!(want_axis && keep == L.AX)) {  // (never with the overlap: it needs keep == L.AX)
    for (int e = t; e < D * D; e += NT) L.AX[(e / D) * LD + e % D] = 0.0;
    if (t < D) L.lam[t] = 0.0;
  }
  if (!want_axis) {
    __syncthreads();
    return true;
  }
  if (D == 1) {
    __syncthreads();
    if (t == 0) {
      const double c = keep ? keep[0] : cov[0];
      if (keep) {
        L.V[0] = c;  // (the copy moves out of L.AX)
      }
      L.AX[0] = sqrt(c);
      L.lam[0] = c;
    }
    if (cov_keep && keep) *cov_keep = L.V;
    __syncthreads();
    return true;
  }
  // ---- dominant eigenvector by repeated squaring ----
  double* P = L.A;
  double* Q = L.V;
  double trP;
  bool conv = false;
  int it0 = 0;
  if (ovl) {
    // what wavefronts 2, 3 have done beside the sweeps; the covariance comes back into L.A (free since the inverse was
    // symmetrised into L.AM) from the node's global working copy
    trP = s_ovl_tr;
    conv = s_ovl[0] != 0;
    it0 = s_ovl[1];
    P = s_ovl[2] ? L.V : L.AX;
    Q = s_ovl[2] ? L.AX : L.V;
    if (conv)
      for (int e = t; e < D * D; e += NT) L.A[(e / D) * LD + e % D] = cov[(e / D) * LD + e % D];
    __syncthreads();
  } else {
    {
      const double s0 = ldexp(1.0, -(ilogb(tr_cov) + 1));  // trace in [1/2, 1)
      if (keep)
        for (int e = t; e < D * D; e += NT) P[(e / D) * LD + e % D] = keep[(e / D) * LD + e % D] * s0;
      else
        for (int e = t; e < D * D; e += NT) P[(e / D) * LD + e % D] = cov[(e / D) * LD + e % D] * s0;
    }
    __syncthreads();
    trP = wave_trace(P, D, LD);
  }
  for (int it = it0; it < kFastSquarings && !conv; ++it) {
    // the product is scaled by a power of two (exact) chosen from tr(P)^2, the upper bound of its
    // trace: tr(P^2) / tr(P)^2 = r in [1/D, 1] is the sum of the squared eigenvalue weights of P, so the
    // stored trace stays within [1/(4D), 1) without a pass of its own
    const double sc = ldexp(1.0, -(ilogb(trP * trP) + 1));
    sym_square<NT>(P, Q, D, LD, sc);
    __syncthreads();
    const double trQ = wave_trace(Q, D, LD);
    const double r = trQ / (trP * trP * sc);
    // 1 - r ~ 2 w_2: once w_2(P) is below ~5e-9 the product just formed has w_2^2 < 1e-16
    conv = 1.0 - r < 1e-8;
    trP = trQ;
    double* tmp = P;
    P = Q;
    Q = tmp;
    if (ovl && conv) {  // (the covariance back into L.A, as above)
      for (int e = t; e < D * D; e += NT) L.A[(e / D) * LD + e % D] = cov[(e / D) * LD + e % D];
      __syncthreads();
    }
  }
  if (!conv) return false;
  PH_ADD(14);
  // column with the largest diagonal entry (v_j^2); normalise; canonical sign (largest component > 0)
  __shared__ double s_v[64];
  if (t < 64) {
    const double pdiag = P[(t < D ? t : 0) * (LD + 1)];
    double best = t < D ? pdiag : -1.0;
    int bi = t < D ? t : 0;
    auto take_max = [&](double ob, int oi) {  // larger value, lower index on ties
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    };
    take_max(xor_lane<32>(best), xor_lane_i32<32>(bi));
    take_max(xor_lane<16>(best), xor_lane_i32<16>(bi));
    take_max(xor_lane<8>(best), xor_lane_i32<8>(bi));
    take_max(xor_lane<4>(best), xor_lane_i32<4>(bi));
    take_max(xor_lane<2>(best), xor_lane_i32<2>(bi));
    take_max(xor_lane<1>(best), xor_lane_i32<1>(bi));
    const double x = sel_ld(P, t * LD + bi, t < D);
    const double nn = wave_sum(x * x);
    double ax = fabs(x);
    int ai = t;
    auto take_abs = [&](double oa, int oi) {
      if (oa > ax || (oa == ax && oi < ai)) {
        ax = oa;
        ai = oi;
      }
    };
    take_abs(xor_lane<32>(ax), xor_lane_i32<32>(ai));
    take_abs(xor_lane<16>(ax), xor_lane_i32<16>(ai));
    take_abs(xor_lane<8>(ax), xor_lane_i32<8>(ai));
    take_abs(xor_lane<4>(ax), xor_lane_i32<4>(ai));
    take_abs(xor_lane<2>(ax), xor_lane_i32<2>(ai));
    take_abs(xor_lane<1>(ax), xor_lane_i32<1>(ai));
    const double xs = __shfl(x, ai);
    const double v = x * (xs < 0.0 ? -1.0 : 1.0) / sqrt(nn);
    if (t < D) s_v[t] = v;
  }
  __syncthreads();
  // lam_max = v^T cov v
  if (keep != L.AX) {
    if (t < 64) {
      double y = 0.0;
      if (t < D) {
        if (keep)
          for (int k = 0; k < D; ++k) y = fma(keep[t * LD + k], s_v[k], y);
        else
          for (int k = 0; k < D; ++k) y = fma(cov[t * LD + k], s_v[k], y);
      }
      const double q = wave_sum(t < D ? y * s_v[t] : 0.0);
      if (t < D) L.AX[t * LD] = s_v[t] * sqrt(q);
      if (t == 0) L.lam[0] = q;
    }
    __syncthreads();
    PH_ADD(15);
    return L.lam[0] > 0.0 && isfinite(L.lam[0]);
  }
  // the copy sits in L.AX: wave 0 forms the quotient from it, then every thread moves its entries of the copy to the
  // squaring buffer that is free now (Q) and writes the axis matrix -- column 0 = sqrt(lam_max) v, zero elsewhere: the
  // same values as above
  __shared__ double s_q;
  const double* cvm = ovl ? L.A : L.AX;  // where the covariance sits now
  if (t < 64) {
    double y = 0.0;
    if (t < D)
      for (int k = 0; k < D; ++k) y = fma(cvm[t * LD + k], s_v[k], y);
    const double q = wave_sum(t < D ? y * s_v[t] : 0.0);
    if (t == 0) s_q = q;
  }
  __syncthreads();
  {
    const double q = s_q, rq = sqrt(q);
    for (int e = t; e < D * D; e += NT) {
      const int i = e / D, jj = e - i * D, o = i * LD + jj;
      if (!ovl) Q[o] = L.AX[o];
      L.AX[o] = jj == 0 ? s_v[i] * rq : 0.0;
    }
    if (t < D) L.lam[t] = t == 0 ? q : 0.0;
    if (cov_keep) *cov_keep = ovl ? L.A : Q;
    __syncthreads();
    PH_ADD(15);
    return q > 0.0 && isfinite(q);
  }
}

// (round 6) ellipsoid_rescale + ellipsoid_store_fast in one pass over the record: the rescaled covariance is formed from
// the LDS copy spd_fast kept (no read of the global working copy, no read-modify-write of it in front of a second
// read), precision matrix and axis are scaled on their way out.  The same products and quotients: the same bits.
template <int NT = kThreads>
__device__ __forceinline__ int ellipsoid_finish_fast(const Lds& L, const RebuildArgs& a, double* es, double* cov_g,
                                                     const double* cov_keep, double fmx, double logdet,
                                                     double* logvol_out) {
  const int D = a.d, t = threadIdx.x, LD = L.LD, DD = D * D;
  if (!isfinite(logdet)) return DH_ERR_VALUE;
  const double lim = 1.0 - kRoundDelta;
  const bool sc = fmx > lim;
  const double mult = fmx / lim, rt = sqrt(mult);
  if (t < D) {
    st_c(L, es + t, L.mean[t]);
    double lam = L.lam[t];
    if (sc) lam *= mult;
    st_c(L, es + D + 3 * DD + t, sqrt(lam));
  }
  for (int e = t; e < DD; e += NT) {
    const int i = e / D, j = e - i * D, o = i * LD + j;
    double c = cov_keep[o], am = L.AM[o], ax = L.AX[o];
    if (sc) {
      c *= mult;
      am /= mult;
      ax *= rt;
      cov_g[o] = c;
    }
    st_c(L, es + D + e, c);
    st_c(L, es + D + DD + e, am);
    st_c(L, es + D + 2 * DD + e, ax);
  }
  *logvol_out = a.prefactor + 0.5 * logdet;
  return DH_OK;
}

// enlarge the ellipsoid so that the outermost point sits at 1 - ROUND_DELTA (bounding.py:1438-1448)
template <int NT = kThreads>
__device__ __forceinline__ void ellipsoid_rescale(const Lds& L, double* cov_g, int D, double fmx) {
  const int t = threadIdx.x, LD = L.LD;
  const double lim = 1.0 - kRoundDelta;
  if (fmx > lim) {
    const double mult = fmx / lim;
    const double rt = sqrt(mult);
    for (int e = t; e < D * D; e += NT) {
      const int i = e / D, j = e % D;
      cov_g[i * LD + j] *= mult;
      L.AM[i * LD + j] /= mult;
      L.AX[i * LD + j] *= rt;
    }
    __syncthreads();
    if (t < D) L.lam[t] *= mult;
    __syncthreads();
  }
}

// Ellipsoid.__init__ (bounding.py:201-240): eigenvalues must be positive; log-volume; the
// record ctr | cov | am | axes | axlens -> es
__device__ __forceinline__ int ellipsoid_store(const Lds& L, const RebuildArgs& a, double* es, const double* cov_g,
                                               double* logvol_out) {
  const int D = a.d, t = threadIdx.x, LD = L.LD;
  bool ok = true;
  double slog = 0.0;
  for (int k = 0; k < D; ++k) {
    const double l = L.lam[k];
    if (!(l > 0.0) || !isfinite(l)) ok = false;
    slog += log(l);
  }
  if (!ok) return DH_ERR_VALUE;
  const double logvol = a.prefactor + 0.5 * slog;
  const int DD = D * D;
  if (t < D) {
    st_c(L, es + t, L.mean[t]);
    st_c(L, es + D + 3 * DD + t, sqrt(L.lam[t]));
  }
  for (int e = t; e < DD; e += kThreads) {
    const int i = e / D, j = e % D;
    st_c(L, es + D + e, cov_g[i * LD + j]);
    st_c(L, es + D + DD + e, L.AM[i * LD + j]);
    st_c(L, es + D + 2 * DD + e, L.AX[i * LD + j]);
  }
  __syncthreads();
  *logvol_out = logvol;
  return DH_OK;
}

// record of an eigen-free node: ctr | cov | am | axes (column 0 = major axis, rest 0) | axlens
// (entry 0 = its length, rest 0: k_split's argmax picks column 0).  ln vol = prefactor + ln det / 2.
template <int NT = kThreads>
__device__ __forceinline__ int ellipsoid_store_fast(const Lds& L, const RebuildArgs& a, double* es,
                                                    const double* cov_g, double logdet, double* logvol_out) {
  const int D = a.d, t = threadIdx.x, LD = L.LD;
  if (!isfinite(logdet)) return DH_ERR_VALUE;
  const int DD = D * D;
  if (t < D) {
    st_c(L, es + t, L.mean[t]);
    st_c(L, es + D + 3 * DD + t, sqrt(L.lam[t]));
  }
  for (int e = t; e < DD; e += NT) {
    const int i = e / D, j = e % D;
    st_c(L, es + D + e, cov_g[i * LD + j]);
    st_c(L, es + D + DD + e, L.AM[i * LD + j]);
    st_c(L, es + D + 2 * DD + e, L.AX[i * LD + j]);
  }
  __syncthreads();
  *logvol_out = a.prefactor + 0.5 * logdet;
  return DH_OK;
}

// bounding_ellipsoid (bounding.py:1387-1461) of the node segment; writes the
// ellipsoid record to `es` (global).  Returns 0 or a DH_ERR code (uniform).
constexpr int kFullRecord = 1;  // node_ellipsoid<true>: done, but by the reference route (complete eigen record)

// FAST = true: the eigen-free path; a node it does not apply to takes the reference's route in place with
//              the compact wave-level solver (returns kFullRecord);
// FAST = false: the reference's route (improve_covar_mat with a full eigh per trial) for every node.
// CHILD = true (round 6; the level kernel k_ell<false>): the node is a k-means child -- its mean is in the record, so the
// mean pass and the root's tile-ordered covariance are not even compiled in -- and a node the eigen-free path declines
// is handed to the work-queue tail (kDeferred) instead of taking the reference's route in place: k_ell<false> carried
// the wave-level Jacobi, the 100-trial loop and a second Mahalanobis pass for a path that tree nodes of real live sets
// almost never take, and paid for them in registers (256 + 40 spilled).  The tail's workers run the full routine.
constexpr int kDeferred = 3;
template <bool FAST, bool CHILD = false>
__device__ __forceinline__ int node_ellipsoid(const Lds& L, const RebuildArgs& a, const double* pts, const int* perm,
                                              int start, int count, double* es, double* cov_g, double* logvol_out,
                                              double* fmax_out, bool have_mean = false) {
  const int D = a.d, t = threadIdx.x, LD = L.LD;
  if (count == 1) return DH_ERR_VALUE;
  PH_T0();
  if (CHILD || have_mean) {  // left in the record by k_split (the final k-means centroid of this cluster)
    if (t < D) L.mean[t] = ld_c(L, es + t);
    __syncthreads();
  } else {
    node_mean(L, pts, perm, start, count, D);
  }
  PH_ADD(0);
  int fused = 0;  // 1 / 2: the fused fold has filed the covariance everywhere (finite / not finite)
  node_cov(L, pts, perm, start, count, D, CHILD ? false : !have_mean, FAST ? cov_g : nullptr, &fused);  // no mean handed down = the root
  PH_ADD(1);
  // cov_g: this node's D x LD working covariance (global scratch, L2 resident)
  if (!fused) {
    for (int e = t; e < D * D; e += kThreads) cov_g[(e / D) * LD + e % D] = L.A[(e / D) * LD + e % D];
    __syncthreads();
  }
  if constexpr (FAST) {
    // eigen-free path: good_mat is certain, so the reference's loop ends after its first pass
    double logdet = 0.0;
    const double* cov_keep = nullptr;
    if (fused != 2 && spd_fast(L, cov_g, D, a.mode == 0 && count >= 4 * D, &logdet, &cov_keep, fused == 1)) {
      const double fmx = node_fmax(L, pts, perm, start, count, D);
      PH_ADD(3);
      if (fmx > 1.0 - kRoundDelta) logdet += (double)D * log(fmx / (1.0 - kRoundDelta));
      *fmax_out = fmin(fmx, 1.0 - kRoundDelta);  // the quadratic forms scale with am: fmx / mult
      if (cov_keep != nullptr) return ellipsoid_finish_fast(L, a, es, cov_g, cov_keep, fmx, logdet, logvol_out);
      ellipsoid_rescale(L, cov_g, D, fmx);
      return ellipsoid_store_fast(L, a, es, cov_g, logdet, logvol_out);
    }
    // the eigen-free path does not apply (not positive definite, condition bound >= 1e7, degenerate
    // leading eigenvalues): the reference's route, right here.  (round 6: with the whole-workgroup solver.  The
    // wave-level Jacobi used to stand here -- a fraction of the registers, for a "rare" path -- and took 1.8 ms per
    // node: in a C2 run the deep levels of three early rebuilds produce 16-175 leaves of ~55 points whose covariance
    // bound is above 1e7, and the work-queue tail those are handed to ran 1.84 ms instead of 5 us: 5.5 ms of the
    // loop's 135.  The level kernel hands such nodes on (CHILD: kDeferred), so its registers do not pay for this.)
    if constexpr (CHILD) return kDeferred;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
      const bool good = regularize<false>(L, cov_g, D);
      const double fmx = node_fmax(L, pts, perm, start, count, D);
      if (pass == 0) ellipsoid_rescale(L, cov_g, D, fmx);
      if (pass == 1 && fmx >= 1.0) return DH_ERR_CONTAIN;
      *fmax_out = pass == 0 ? fmin(fmx, 1.0 - kRoundDelta) : fmx;
      if (good) break;
    }
    const int rc = ellipsoid_store(L, a, es, cov_g, logvol_out);
    return rc == DH_OK ? kFullRecord : rc;
  } else {
    for (int pass = 0; pass < 2; ++pass) {
      const bool good = regularize(L, cov_g, D);
      PH_ADD(2);
      const double fmx = node_fmax(L, pts, perm, start, count, D);
      PH_ADD(3);
      if (pass == 0) ellipsoid_rescale(L, cov_g, D, fmx);
      if (pass == 1 && fmx >= 1.0) return DH_ERR_CONTAIN;
      *fmax_out = pass == 0 ? fmin(fmx, 1.0 - kRoundDelta) : fmx;
      if (good) break;
    }
    return ellipsoid_store(L, a, es, cov_g, logvol_out);
  }
}

// ---- k-means (k = 2) + stable partition of a node, cooperatively by its parts -----------
// kmeans2(points/scale, seeds/scale, iter=10, minit='matrix') (bounding.py:1510-1514; the
// scipy.cluster.vq.kmeans2 loop).  A node of c points is handled by np = ceil(c / TP)
// workgroups; part q keeps points [q TP, (q+1) TP) of the node resident in LDS for all
// ten iterations (the old form re-gathered every tile from HBM/L2 in every iteration).
// Per iteration a part assigns its points (same distance arithmetic as vq), reduces the
// per-cluster sums on the matrix cores (sums = Labels^T X), publishes them, meets the
// other parts at a device-scope barrier, and every part forms the new centroids from the
// partials in part order (deterministic).

// device-scope barrier among the np parts of one node; returns false on timeout.
// No cache-wide release/acquire fences (an agent-scope fence writes back / invalidates the
// whole XCD L2, and with hundreds of parts doing that 11 times each the L2 never holds
// anything): everything the parts exchange is written with agent-scope stores (write-through)
// and read with agent-scope loads (bypass), so the barrier only has to order, not to flush.
// Ordering needs every wave's stores ACKNOWLEDGED before thread 0 announces the arrival: s_barrier alone
// does not wait for them (the compiler emits `s_waitcnt lgkmcnt(0); s_barrier` for __syncthreads(); a
// store still in flight to another L2 channel can land after the counter's atomic and a partner then
// reads the previous iteration's partial -- seen as 3 differing results in 2 700 eggbox rebuilds once
// 128-point parts made multi-part nodes common), hence the explicit vmcnt(0).
#ifndef DH_BAR_SLEEP
#define DH_BAR_SLEEP 4
#endif
__device__ __forceinline__ void drain_stores() {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt(0), expcnt / lgkmcnt untouched
}

// ---- tagged exchange of the k-means partials (round 6) -----------------------------------------------------------
// An iteration of a multi-part node used to cost four device-scope round trips in a row (13 000 cycles with 16 parts:
// the largest item of the top levels): partials stored, stores acknowledged (vmcnt 0), arrival counted by a returning
// atomic, counter polled, partials fetched.  Now every partial travels as ONE 16-byte store (value, tag) -- tag =
// (rebuild epoch, level, iteration), unique for the slot's whole life -- and a reader polls the words it needs until
// they carry the tag of the iteration: no counter, no acknowledgement, the data is its own signal; an iteration costs
// one store propagation and the polling loads that overlap it.  16-byte global accesses are single requests (not
// torn); agent scope (sc1: write through / bypass, as the compiler emits for the 8-byte __hip_atomic forms).
typedef double dh_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_agent16(double* p, double v, unsigned long long tag) {
  dh_d2 x;
  x.x = v;
  x.y = __longlong_as_double((long long)tag);
  // (s_nop: a store of more than 8 bytes must not be followed at once by a write to its data registers -- the compiler's
  // hazard recognizer inserts the wait state for its own stores, it does not look inside an asm statement.  Without it
  // a build whose next instruction happened to clear v[2:3] published zeros: found with the spilling register budgets)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// eight 16-byte agent-scope loads in flight together, one wait
__device__ __forceinline__ void ld_agent16x8(const double* const (&p)[8], dh_d2 (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}

// column `col` of np partners' rows (rows of `stride` pairs), every word awaited until it carries `tag`: the sum in
// partner order (MAXOP: the maximum) -- the tagged exchange as a reduction.  *ok = false if a partner never arrives.
template <bool MAXOP = false>
__device__ __forceinline__ double tagged_partner_reduce(const double* base, size_t stride, int col, int np,
                                                        unsigned long long tag, bool* ok) {
  double acc = MAXOP ? -INFINITY : 0.0;
  for (int pp0 = 0; pp0 < np && *ok; pp0 += 8) {
    const double* ptr[8];
    dh_d2 v[8];
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) ptr[uu] = base + ((size_t)min(pp0 + uu, np - 1) * stride + col) * 2;
    for (int spins = 0;; ++spins) {
      ld_agent16x8(ptr, v);
      bool all = true;
#pragma unroll
      for (int uu = 0; uu < 8; ++uu) all = all && (unsigned long long)__double_as_longlong(v[uu].y) == tag;
      if (all) break;
      if (spins > (1 << 18)) {
        *ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      if (MAXOP)
        acc = pp0 + uu < np ? fmax(acc, v[uu].x) : acc;
      else
        acc += pp0 + uu < np ? v[uu].x : 0.0;
    }
  }
  return acc;
}

__device__ __forceinline__ bool parts_barrier(int* bar, int target) {
  __shared__ int ok_flag;
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    long long spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(DH_BAR_SLEEP);
      if (++spins > (1ll << 20)) {  // partners never arrived (would otherwise hang the device)
        ok = 0;
        break;
      }
    }
    ok_flag = ok;
  }
  __syncthreads();
  return ok_flag != 0;
}

// The label sums of one wave's 64 points (see node_kmeans_part): two accumulator chains per dimension block, groups of
// four points alternating between them -- the arithmetic of rounds 3-5, bit for bit.  What round 6 changed is how the
// operands arrive.  The old loop asked `point valid && dimension valid ? tile[..] : 0` and `labels[p] == cluster` of
// LDS for every operand: every one of those loads became a block of its own under a saved exec mask with `s_waitcnt 0`
// behind it -- 48 serialised LDS round trips, 4 100 cycles per Lloyd iteration.  Now:
//   * the indicators come from the wave's own label ballots (`sel`: bit b set = point b of this wave carries this
//     lane's indicator row), no label array in LDS;
//   * rows beyond the part's points are ZERO in the tile (node_kmeans_part fills them after staging), so a row load
//     needs no predicate; columns beyond D read whatever follows in LDS -- they only ever reach output columns >= D,
//     which nobody reads (a 4x4x4 block's column j depends on B's column j alone);
//   * four groups of eight points are straight-line code (all loads in flight before the first product), twice.
template <bool NB3>
__device__ __forceinline__ void kmeans_label_sums(const Lds& L, unsigned long long sel, double& o0, double& o1, double& o2) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, LD = L.LD;
  const int lj = lane & 15, lk = lane >> 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
  const unsigned long long mine = sel >> lk;  // bit 8 g (+ 4): this lane's indicator for the (second) point of group g
  const double* base = L.tile + (w * 64 + lk) * LD + lj;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int g8 = h * 4 + g;
      const double ind = ((mine >> (8 * g8)) & 1ull) ? 1.0 : 0.0;
      const double indb = ((mine >> (8 * g8 + 4)) & 1ull) ? 1.0 : 0.0;
      const double* row = base + g8 * 8 * LD;
      const double* rowb = row + 4 * LD;
      a0 = DH_MFMA_F64_4X4(ind, row[0], a0);
      b0 = DH_MFMA_F64_4X4(indb, rowb[0], b0);
      a1 = DH_MFMA_F64_4X4(ind, row[16], a1);
      b1 = DH_MFMA_F64_4X4(indb, rowb[16], b1);
      if constexpr (NB3) {
        a2 = DH_MFMA_F64_4X4(ind, row[32], a2);
        b2 = DH_MFMA_F64_4X4(indb, rowb[32], b2);
      }
    }
  }
  o0 = a0 + b0;
  o1 = a1 + b1;
  o2 = a2 + b2;
}

// lane l's value of a double, to every lane (l uniform): two v_readlane_b32 -- no LDS round trip
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// One part's share of the k-means + partition of node [start, start+count).  q = part
// index, np = number of parts, kp = this node's partial-sum slots (2 parities x np x KP),
// bar = its barrier counter.  Returns n0 (size of cluster 0) or -1 on a barrier timeout;
// on return perm[start + q TP ...] holds the partitioned order if the split is viable.
__device__ int node_kmeans_part(const Lds& L, const double* pts, int* perm, int* perm2, int start, int count,
                                int D, const double* es, int q, int np, double* kp0, double* kp1, int* bar,
                                int min_size, unsigned long long tag0, int lvl = 0) {
  LV_T0();
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, LD = L.LD;
  const int DD = D * D, KP = 2 * D + 2;
  const int s0 = start + q * L.TP, cnt = min(L.TP, count - q * L.TP);
  // seeds: major-axis endpoints ctr -/+ axes[:, argmax(axlens)] (bounding.py:278-284)
  if (t < D) L.sums[t] = ld_c(L, es + D + 3 * DD + t);  // axis lengths: one parallel fetch, then a scan in LDS
  __syncthreads();
  if (t == 0) {
    L.ri[303] = 0;  // (set if a partner's partials never arrive)
    int best = 0;
    double bl = L.sums[0];
    for (int k = 1; k < D; ++k)
      if (L.sums[k] > bl) {
        bl = L.sums[k];
        best = k;
      }
    L.ri[301] = best;
  }
  stage_tile(L, pts, perm, s0, cnt, D, 0);  // pts = points / scale (pre-divided); resident from here on
  const int kbest = L.ri[301];
  if (t < D) {
    const double v = ld_c(L, es + D + 2 * DD + t * D + kbest), ct = ld_c(L, es + t);
    L.cen[t] = (ct - v) / L.scale[t];
    L.cen[D + t] = (ct + v) / L.scale[t];
  }
  __syncthreads();
  const int nb = (D + 15) >> 4;
  const int lj = lane & 15, lk = lane >> 4;
  int lb = 0, lb_prev = -1, n0 = 0, c0_tile = 0, last_it = 9;
  LV_ADD(lvl, 0);
  LV_SET(lvl, 5, np);
  LV_SET(lvl, 6, count);
  PH_T0();
  // (round 6) rows [cnt, 64 ceil(cnt / 64)) of the tile are zeroed once: the label sums then load rows without a
  // predicate (an indicator of 0 times a stale NaN would still be NaN)
  {
    const int rup = min(L.TP, (cnt + 63) & ~63);
    const int jz = t & (L.DP - 1), pz0 = cnt + (t >> L.DPlog), pzs = kThreads >> L.DPlog;
    if (jz < D)
      for (int pz = pz0; pz < rup; pz += pzs) L.tile[pz * LD + jz] = 0.0;
    __syncthreads();
  }
  unsigned long long m0 = 0ull, m1 = 0ull;  // this wave's label ballots (valid points only)
  for (int it = 0; it < 10; ++it) {
    // vq: nearest centroid, strict '<' so the lower index wins ties.  (round 6) The centroids ride in the lanes of two
    // registers (lane j: coordinate j) and reach the distance loop through v_readlane: two LDS loads per wave and
    // iteration instead of two per dimension and point -- with four to five parts per CU the LDS pipe was half of
    // an iteration's time at 64 runs.  Same differences, same fma chains: the same bits.
    const bool has = t < cnt;
    if (w * 64 < cnt) {  // (uniform per wave)
      const double c0l = L.cen[lane < D ? lane : 0], c1l = L.cen[D + (lane < D ? lane : 0)];
      const double* x = L.tile + (has ? t : 0) * LD;
      double d0 = 0.0, d1 = 0.0;
      // eight coordinates of the point fetched together (clamped offsets: unconditional loads), then their eight
      // steps of the two chains -- the compiler does not unroll the loop around the lane reads by itself, and one
      // LDS round trip per dimension was 4 000 cycles an iteration
      for (int j0 = 0; j0 < D; j0 += 8) {
        double xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = x[min(j0 + u, D - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < D) {  // (uniform)
            const double e0 = xv[u] - readlane_f64(c0l, j0 + u), e1 = xv[u] - readlane_f64(c1l, j0 + u);
            d0 = fma(e0, e0, d0);
            d1 = fma(e1, e1, d1);
          }
      }
      lb = d1 < d0 ? 1 : 0;
    }
    LV_ADD(lvl, 8);
    {
      // labels that moved since the last iteration (per wave, summed after the barrier): none anywhere in the node
      // = a fixed point of the Lloyd iteration, the remaining iterations would reproduce this one bit for bit
      const unsigned long long mv = __ballot(has && lb != lb_prev);
      m1 = __ballot(has && lb == 1);
      m0 = __ballot(has && lb == 0);
      if (lane == 0) {
        L.ri[260 + w] = __popcll(mv);
        L.ri[264 + w] = __popcll(m0);
      }
      lb_prev = lb;
    }
    __syncthreads();  // (one barrier: __syncthreads_count is three)
    c0_tile = L.ri[264] + L.ri[265] + L.ri[266] + L.ri[267];
    const int ch_tile = L.ri[260] + L.ri[261] + L.ri[262] + L.ri[263];
    PH_ADD(10);
    LV_ADD(lvl, 9);
    // update_cluster_means: per-cluster sums = Labels^T X on the matrix cores; wave w
    // contracts its 64 points (rows 0/1 of the 16-row A operand are the two indicators)
    // Only two of a 16x16x4 tile's sixteen rows would carry indicators, and on gfx950 that
    // instruction costs 64 cycles (tools/micro/mfma_f64_shapes.hip); the 4x4x4 form (17-24 cycles:
    // four independent 4x4x4 products) takes the indicators as its 4 rows and 4 x 4 dimensions as the
    // columns of its four blocks.  Operand lanes: A row = lane & 3, k = lane >> 4 (any block);
    // B column = dimension lane & 15, k = lane >> 4; D row = lane >> 4, dimension lane & 15.
    // two independent accumulator chains (groups of 4 points alternate between them): a dependent
    // chain of 16 x 3 matrix instructions per wave was most of this phase
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (w * 64 < cnt) {
      const int li = lane & 3;
      const unsigned long long sel = li == 0 ? m0 : li == 1 ? m1 : 0ull;
      if (nb > 2)
        kmeans_label_sums<true>(L, sel, a0, a1, a2);
      else
        kmeans_label_sums<false>(L, sel, a0, a1, a2);
    }
    // result rows 0 and 1 (the two clusters) sit in lanes 0..15 and 16..31
    if (lane < 32) {
      double* o = L.kred + (w * 2 + lk) * 48;
      o[lj] = a0;
      if (nb > 1) o[16 + lj] = a1;
      if (nb > 2) o[32 + lj] = a2;
    }
    LV_ADD(lvl, 10);
    __syncthreads();
    PH_ADD(11);
    LV_ADD(lvl, 11);
    double* kp = (it & 1) ? kp1 : kp0;
    if (t < 2 * D) {
      const int c = t >= D ? 1 : 0, j = t - c * D;
      // (grouped like the parts of the level kernels when one 256-point tile stands in for two 128-point
      // parts -- L.KG < 4 -- so that a node's sums do not depend on the tile that held it)
      double sum = 0.0;
      for (int g0 = 0; g0 < kThreads / 64; g0 += L.KG) {
        double gs = 0.0;
        for (int wv = g0; wv < g0 + L.KG; ++wv) gs += L.kred[(wv * 2 + c) * 48 + j];
        sum += gs;
      }
      L.sums[c * D + j] = sum;
      if (np == 1) {
        // a node that is one part: the thread that holds a cluster sum forms the centroid entry right away (the
        // same quotient as below) -- one barrier and one LDS round trip less per iteration
        const int nc = c ? count - c0_tile : c0_tile;
        if (nc > 0) L.cen[c * D + j] = sum / (double)nc;  // empty cluster keeps its centroid
      }
    }
    n0 = c0_tile;
    LV_ADD(lvl, 12);
    if (np > 1) {
      // publish this part's row (columns 0 .. 2D-1: the cluster sums, 2D: the label-0 count, 2D + 1: labels that moved),
      // then fetch every partner's -- the threads that file a column poll it, summed in part order (deterministic)
      const unsigned long long tag = tag0 + (unsigned long long)(it + 1);
      if (t < KP) {
        const double mine = t < 2 * D ? L.sums[t] : t == 2 * D ? (double)c0_tile : (double)ch_tile;  // (own entry: written by this thread)
        st_agent16(kp + ((size_t)q * KP + t) * 2, mine, tag);
        double sum = 0.0;
        bool ok = true;
        for (int pp0 = 0; pp0 < np && ok; pp0 += 8) {
          const double* ptr[8];
          dh_d2 v[8];
#pragma unroll
          for (int uu = 0; uu < 8; ++uu) ptr[uu] = kp + ((size_t)min(pp0 + uu, np - 1) * KP + t) * 2;
          for (int spins = 0;; ++spins) {
            ld_agent16x8(ptr, v);
            bool all = true;
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) all = all && (unsigned long long)__double_as_longlong(v[uu].y) == tag;
            if (all) break;
            if (spins > (1 << 18)) {  // a partner never arrived (would otherwise hang the device)
              ok = false;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
#pragma unroll
          for (int uu = 0; uu < 8; ++uu) sum += pp0 + uu < np ? v[uu].x : 0.0;
        }
        if (!ok) L.ri[303] = 1;
        L.sums[t] = sum;
      }
      __syncthreads();
      if (L.ri[303]) return -1;
      n0 = (int)L.sums[2 * D];
    }
    LV_ADD(lvl, 13);
    const int moved = np > 1 ? (int)L.sums[2 * D + 1] : ch_tile;
    if (np > 1) {
      const int n1 = count - n0;
      if (t < D) {
        if (n0 > 0) L.cen[t] = L.sums[t] / (double)n0;  // empty cluster keeps its centroid
        if (n1 > 0) L.cen[D + t] = L.sums[D + t] / (double)n1;
      }
    }
    __syncthreads();
    PH_ADD(12);
    LV_ADD(lvl, 14);
    if (moved == 0) {  // (never at it = 0: every label counts as moved there)
      last_it = it;
      break;
    }
  }
  LV_ADD(lvl, 1);
  LV_SET(lvl, 4, last_it + 1);
  if (min(n0, count - n0) < min_size) return n0;  // split rejected (:1521-1522): no partition needed
  // ---- stable partition by label (label 0 first) ----
  // ranks inside the tile from wave ballots; offsets of this part from the partners' counts
  const bool valid = t < cnt;
  const unsigned long long pm0 = __ballot(valid && lb == 0);
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (lane == 0) L.ri[256 + w] = __popcll(pm0);
  __syncthreads();
  int before0 = 0;
  for (int wv = 0; wv < w; ++wv) before0 += L.ri[256 + wv];
  const int rank0 = before0 + __popcll(pm0 & lt);
  int off0 = 0, off1 = n0;
  if (np > 1) {
    const double* kp = (last_it & 1) ? kp1 : kp0;  // the last iteration run
    for (int pp0 = 0; pp0 < q; pp0 += 8) {
      double part[8];
#pragma unroll
      for (int uu = 0; uu < 8; ++uu)
      {
        const double pv8 = ld_agent(kp + ((size_t)min(pp0 + uu, q - 1) * KP + 2 * D) * 2);  // (the value half of the pair)
        part[uu] = pp0 + uu < q ? pv8 : 0.0;
      }
#pragma unroll
      for (int uu = 0; uu < 8; ++uu)
        if (pp0 + uu < q) {
          off0 += (int)part[uu];
          off1 += L.TP - (int)part[uu];  // parts before q are full tiles
        }
    }
  }
  if (valid) {
    const int pos = lb == 0 ? off0 + rank0 : off1 + (t - rank0);
    const int src = ld_ci(L, perm + s0 + t);
    if (np > 1 || L.coh)
      st_agent_i(perm2 + start + pos, src);
    else
      perm2[start + pos] = src;
  }
  if (np > 1) {
    if (!parts_barrier(bar, np)) return -1;  // (the node's only counted barrier: the iterations exchange tagged words)
    if (valid) st_ci(L, perm + s0 + t, ld_agent_i(perm2 + s0 + t));
  } else if (L.coh) {
    drain_stores();
    __syncthreads();
    if (valid) st_agent_i(perm + s0 + t, ld_agent_i(perm2 + s0 + t));
  } else {
    __threadfence_block();
    __syncthreads();
    if (valid) perm[s0 + t] = perm2[s0 + t];
  }
  LV_ADD(lvl, 2);
  return n0;
}

__device__ __forceinline__ double logaddexp_d(double x, double y) {
  // np.logaddexp
  if (x == y) return x + 0.6931471805599453;
  const double d = x - y;
  if (d > 0) return x + log1p(exp(-d));
  if (d <= 0) return y + log1p(exp(d));
  return x + y;  // NaN
}

// ---- the rebuild as a level-synchronous kernel pipeline ----------------------
//   k_root_parts (grid = runs x ceil(n/TP))  root ellipsoid by cooperating resident parts, per-run scale, worklist seed
//   k_split  (grid = runs * maxw)     one workgroup per splittable node of the
//                                     level: k-means (k=2) + stable partition
//   k_ell    (grid = runs * 2 maxw)   one workgroup per new child: bounding
//                                     ellipsoid; queues it for the next level
//   ... repeated for `levels` levels (idle workgroups exit immediately) ...
//   k_finish (grid = runs)            bottom-up accept test, emit, coverage check
// All nodes of a level -- of every run -- are processed concurrently, so a
// single run is no longer confined to one CU and nothing returns to the host.

// LDS without the Jacobi work buffers (16-byte multiple); they follow at this offset
// when everything fits kLdsLimit, else they overlay the point tile.
constexpr size_t kLdsLimit = 159 * 1024;
// separate Jacobi buffers only while two workgroups still fit one CU's 160 KB
constexpr size_t kLdsSeparate = 79 * 1024;
__host__ __device__ inline size_t rebuild_lds_base_bytes(int D, int TP) {
  const int LD = D | 1;
  const size_t dbl = (size_t)TP * LD + 4 * (size_t)D * LD + 7 * (size_t)D + 2 + kThreads + 128;
  return (dbl * 8 + (320 + (size_t)D + 8) * 4 + 15) & ~(size_t)15;
}

__device__ __forceinline__ void carve(Lds& L, unsigned char* smem, int D, int TP = kThreads) {
  L.LD = D | 1;  // odd leading dimension: conflict-free column walks
  L.TP = TP;
  L.c_pts = nullptr;
  L.c_start = L.c_cnt = L.c_how = -1;
  L.coh = false;
  L.KG = kThreads / 64;
  L.DP = 1;
  L.DPlog = 0;
  while (L.DP < D) {
    L.DP <<= 1;
    ++L.DPlog;
  }
  double* p = (double*)smem;
  L.tile = p;
  p += (size_t)L.TP * L.LD;
  L.A = p;
  p += D * L.LD;
  L.V = p;
  p += D * L.LD;
  L.AM = p;
  p += D * L.LD;
  L.AX = p;
  p += D * L.LD;
  L.mean = p;
  p += D;
  L.scale = p;
  p += D;
  L.lam = p;
  p += D;
  L.cen = p;
  p += 2 * D;
  L.sums = p;
  p += 2 * D + 2;
  L.red = p;
  p += kThreads;
  L.rc = p;
  p += 64;
  L.rs = p;
  p += 64;
  L.kred = L.red;  // 384 doubles = red | rc | rs, none of which the k-means parts use
  L.ibuf = (int*)L.red;
  L.ibuf_cap = 2 * (kThreads + 128);
  L.ri = (int*)p;
  L.perm_sort = L.ri + 320;
  const int P = (D + 1) & ~1;
  L.JLD = P | 1;
  const size_t jdbl = 4 * (size_t)P * L.JLD;
  L.j_alias = rebuild_lds_base_bytes(D, TP) + jdbl * 8 > kLdsSeparate;
  double* jb = L.j_alias ? L.tile : (double*)(smem + rebuild_lds_base_bytes(D, TP));
  L.JA[0] = jb;
  L.JA[1] = jb + (size_t)P * L.JLD;
  L.JV[0] = jb + 2 * (size_t)P * L.JLD;
  L.JV[1] = jb + 3 * (size_t)P * L.JLD;
}

// k_split's own, smaller layout: the resident tile of tps points and what the k-means touches (scale, centroids,
// sums, the partial-sum scratch, the integer scratch) -- 31 KB at D = 25, tps = 128, so that five workgroups share a
// CU.  (With the common layout's 77 KB two did, and a level of the 64-run bench rebuild has 512-1 000 busy parts:
// every level took two rounds of workgroups, 135 us instead of the 65-70 us it takes alone.)
__host__ __device__ inline size_t split_lds_bytes(int D, int TP) {
  const int LD = D | 1;
  return ((((size_t)TP * LD + 5 * (size_t)D + 2 + kThreads + 128) * 8 + (320 + (size_t)D + 8) * 4) + 15) & ~(size_t)15;
}

__device__ __forceinline__ void carve_split(Lds& L, unsigned char* smem, int D, int TP) {
  L.LD = D | 1;
  L.TP = TP;
  L.c_pts = nullptr;
  L.c_start = L.c_cnt = L.c_how = -1;
  L.coh = false;
  L.KG = kThreads / 64;
  L.DP = 1;
  L.DPlog = 0;
  while (L.DP < D) {
    L.DP <<= 1;
    ++L.DPlog;
  }
  double* p = (double*)smem;
  L.tile = p;
  p += (size_t)TP * L.LD;
  L.scale = p;
  p += D;
  L.cen = p;
  p += 2 * D;
  L.sums = p;
  p += 2 * D + 2;
  L.red = p;
  L.rc = p + kThreads;
  L.rs = L.rc + 64;
  L.kred = L.red;
  L.ibuf = (int*)L.red;
  L.ibuf_cap = 2 * (kThreads + 128);
  p += kThreads + 128;
  L.ri = (int*)p;
  L.perm_sort = L.ri + 320;
  L.A = L.V = L.AM = L.AX = L.mean = L.lam = nullptr;
  L.JA[0] = L.JA[1] = L.JV[0] = L.JV[1] = nullptr;
  L.JLD = 0;
  L.j_alias = false;
}

struct RunView {
  const double* pts;
  int* perm;
  int* perm2;
  unsigned char* lab;
  Node* nodes;
  double* estore;
  int* reslist;
  int NS, ES;
};

__device__ __forceinline__ RunView view_of(const RebuildArgs& a, int run, int LD) {
  RunView v;
  const int D = a.d;
  v.pts = a.pts + (size_t)run * a.n * D;
  v.perm = a.perm + (size_t)run * a.n;
  v.perm2 = a.perm2 + (size_t)run * a.n;
  v.lab = a.lab + (size_t)run * a.n;
  v.nodes = a.nodes + (size_t)run * a.max_nodes;
  v.ES = D + 3 * D * D + D;
  v.NS = v.ES + D * LD;
  v.estore = a.estore + (size_t)run * a.max_nodes * v.NS;
  v.reslist = a.reslist + (size_t)run * a.reslist_cap;
  return v;
}

__device__ __forceinline__ void set_status(const RebuildArgs& a, int run, int rc) {
  // first error wins is not needed: any error code marks the run failed
  if (threadIdx.x == 0 && rc != DH_OK) atomicMin(&a.status[run], rc);
}

// queue `node` (count points) for splitting at `level`: one split_list slot + its parts
// ---- work queue of k_tree ------------------------------------------------------------------------
// item = valid bit 63 | kind (bit 62: 0 = ellipsoid of a node, 1 = one k-means part of a node) | run (22 bits) |
// node (24 bits) | part (16 bits).  Producers: count the items as pending FIRST, reserve slots, publish each
// with an agent-scope store -- after everything the consumer will read has been stored (agent scope) and
// acknowledged (drain_stores + barrier in the callers).
constexpr unsigned long long kItemValid = 1ull << 63, kItemSplit = 1ull << 62;
__device__ __forceinline__ unsigned long long tq_item(bool split, int run, int node, int part) {
  return kItemValid | (split ? kItemSplit : 0ull) | ((unsigned long long)run << 40) | ((unsigned long long)node << 16) |
         (unsigned long long)part;
}
// one thread: queue n items (ellipsoid of node0, node0 + 1, ... or parts 0 .. n-1 of node0)
__device__ __forceinline__ bool tq_push(const RebuildArgs& a, bool split, int run, int node0, int n) {
  __hip_atomic_fetch_add(a.tq_ctl + 32, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int s0 = __hip_atomic_fetch_add(a.tq_ctl + 16, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (s0 + n > a.tq_cap) {
    __hip_atomic_fetch_add(a.tq_ctl + 32, -n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicMin(&a.status[run], DH_ERR_NOMEM);
    return false;
  }
  for (int i = 0; i < n; ++i)
    __hip_atomic_store(a.tq_items + s0 + i, split ? tq_item(true, run, node0, i) : tq_item(false, run, node0 + i, 0),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

__device__ __forceinline__ void queue_split(const RebuildArgs& a, int run, int level, int node, int count) {
  if (a.tree || level >= a.tree_from) {
    const int np = (count + a.tps - 1) / a.tps;
    int* nb = a.nbar + ((size_t)run * a.max_nodes + node) * kBarStride;
    int pb = 0;
    if (np > 1) {  // only multi-part nodes exchange partial sums
      pb = atomicAdd(&a.kp_top[run], np);
      if (pb + np > a.kp_cap) {
        atomicMin(&a.status[run], DH_ERR_NOMEM);
        return;
      }
    }
    st_agent_i(nb + 2, pb);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the slot index is acknowledged before the parts are published
    (void)tq_push(a, true, run, node, np);
    return;
  }
  const size_t lp = (size_t)(level & 1) * a.runs + run;
  const int sidx = atomicAdd(&a.nsplit[(size_t)level * a.runs + run], 1);
  const int np = (count + a.tps - 1) / a.tps;
  const int pb = atomicAdd(&a.nparts[(size_t)level * a.runs + run], np);
  if (sidx >= a.maxw || pb + np > a.maxp) {
    atomicMin(&a.status[run], DH_ERR_NOMEM);
    return;
  }
  a.split_list[lp * a.maxw + sidx] = node;
  a.part_base[lp * a.maxw + sidx] = pb;
  for (int qq = 0; qq < np; ++qq) {
    a.part_list[(lp * a.maxp + pb + qq) * 2] = sidx;
    a.part_list[(lp * a.maxp + pb + qq) * 2 + 1] = qq;
  }
}

// ---- the root, cooperatively -----------------------------------------------------------------
// k_root spends most of its time gathering the 8 tiles of a 2000-point live set three times (mean,
// covariance, Mahalanobis maximum) and a fourth and fifth time for std and the scaled copy.
// Here the root is worked on by np = ceil(n / TP) workgroups, each keeping ITS tile resident in LDS
// for the whole kernel: partial column sums -> mean, centred in place, partial Xc^T Xc on the
// matrix cores, Mahalanobis maximum and sum of squares over the own tile; the parts meet at the
// fence-free device-scope barrier of the k-means parts (five times) and exchange their partials
// with agent-scope stores / loads.  Part 0 alone runs the eigensolver (improve_covar_mat) and
// writes the node; if the covariance needed regularising it falls back to the single-workgroup
// routine for the rest (rare path, same code as k_root).
// rootbuf per run: [np x D sums | np x D^2 cov partials | np fmax | np x D squares | D^2 am | 8 flags]
__global__ void __launch_bounds__(kThreads, 2) k_root_parts(RebuildArgs a, int rp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  const int D = a.d, DD = D * D, t = threadIdx.x, run = a.root_run0 + blockIdx.x / rp, q = blockIdx.x % rp;
  const int n = a.n_arr ? a.n_arr[run] : a.n;
  if (a.active && !a.active[run]) return;
  Lds L;
  carve(L, smem, D);
  const int np = (n > 1 && rp > 1) ? (n + L.TP - 1) / L.TP : 1;  // rp == 1: the single-workgroup routine for any n
  if (q >= np) return;
  const RunView v = view_of(a, run, L.LD);
  const int LD = L.LD;
  const int s0 = q * L.TP, cnt = min(L.TP, n - s0);
  // identity permutation: every part its own range (the only part: everything)
  for (int p = s0 + t; p < (np == 1 ? n : s0 + cnt); p += kThreads) v.perm[p] = p;
  __threadfence_block();
  __syncthreads();
  double* es = v.estore;
  double* cov_g = v.estore + v.ES;
  int status = DH_OK;
  double lv = 0.0;
  int root_fast = 0;      // the root's record is the eigen-free form
  double root_logdet = 0.0;
  double root_fmax = INFINITY;  // "unknown": k_finish then runs the coverage pass
  if (np == 1) {
    // small live set: the single-workgroup routine
    if (n <= 1) status = (a.mode == 0) ? DH_ERR_REGION : DH_ERR_VALUE;  // single point
    if (status == DH_OK) {
      if (a.fast) {
        status = node_ellipsoid<true>(L, a, v.pts, v.perm, 0, n, es, cov_g, &lv, &root_fmax);
        root_fast = status == DH_OK;
        if (status == kFullRecord) status = DH_OK;
      } else {
        status = node_ellipsoid<false>(L, a, v.pts, v.perm, 0, n, es, cov_g, &lv, &root_fmax);
      }
    }
  } else {
    // (round 6) The parts exchange their partials as tagged 16-byte words (value, tag = rebuild epoch + phase), polled
    // by whoever needs them: no arrival counters, no store drains -- five counted barriers of ~3 us became four waits
    // of one store propagation each.  And what does not depend on part 0's solve no longer waits for it: the sums of
    // squares travel with the covariance partials, so every part forms the k-means scale and writes its rows of the
    // scaled copy WHILE part 0 inverts the covariance (they were a barrier, a pass and a store behind it).
    // rootbuf per run, in pairs: [rp x D sums | rp x D squares | rp x D^2 cov partials | D^2 am | rp fmax | flag | status]
    double* rb = a.rootbuf + (size_t)run * a.rootbuf_stride;
    double* b_sum = rb;
    double* b_sq = b_sum + (size_t)rp * D * 2;
    double* b_cov = b_sq + (size_t)rp * D * 2;
    double* b_am = b_cov + (size_t)rp * DD * 2;
    double* b_fmx = b_am + (size_t)DD * 2;
    double* b_flag = b_fmx + (size_t)rp * 2;
    double* b_stat = b_flag + 2;
    const unsigned long long tg = ((unsigned long long)a.epoch << 8) | 0x8000000000000000ull;  // (+ phase 1..6)
    __shared__ int s_bad;
    if (t == 0) s_bad = 0;
    const int G = kThreads / D > 0 ? kThreads / D : 1;
    const int j = t % D, g = t / D;
    const bool want_split = a.mode == 0 && n >= 4 * D;
    bool ok = true;
    // ---- mean ----
    stage_tile(L, v.pts, v.perm, s0, cnt, D, 0);
    {
      double acc = 0.0;
      if (t < G * D)
        for (int p = g; p < cnt; p += G) acc += L.tile[p * LD + j];
      L.red[t] = acc;
      __syncthreads();
      if (t < D) {
        double sum = 0.0;
        for (int gg = 0; gg < G; ++gg) sum += L.red[gg * D + t];
        st_agent16(b_sum + ((size_t)q * D + t) * 2, sum, tg + 1);
        L.mean[t] = tagged_partner_reduce(b_sum, D, t, np, tg + 1, &ok) / (double)n;
      }
    }
    __syncthreads();
    // ---- covariance: partial Xc^T Xc of the own tile; sums of squares of the centred tile ----
    stage_tile(L, v.pts, v.perm, s0, cnt, D, 1);  // centred in place
    {
      mfma_acc acc[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[b] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
      tile_cov_accumulate(L, cnt, D, acc);
      __syncthreads();
      cov_fold_waves(L, D, acc);
      for (int e = t; e < DD; e += kThreads) {
        const int i = e / D, k = e - i * D;
        if (i <= k) st_agent16(b_cov + ((size_t)q * DD + e) * 2, L.A[i * LD + k], tg + 2);
      }
    }
    if (want_split) {
      double acc = 0.0;
      if (t < G * D)
        for (int p = g; p < cnt; p += G) {
          const double x = L.tile[p * LD + j];
          acc = fma(x, x, acc);
        }
      __syncthreads();  // (L.red: the fold above is done with it)
      L.red[t] = acc;
      __syncthreads();
      if (t < D) {
        double sum = 0.0;
        for (int gg = 0; gg < G; ++gg) sum += L.red[gg * D + t];
        st_agent16(b_sq + ((size_t)q * D + t) * 2, sum, tg + 2);
      }
    }
    // ---- part 0: eigensolver; publishes the precision matrix (flag 1) or falls back (flag 2) ----
    int flag = 0;
    if (q == 0) {
      for (int e = t; e < DD; e += kThreads) {
        const int i = e / D, k = e - i * D;
        if (i <= k) L.A[i * LD + k] = tagged_partner_reduce(b_cov, DD, e, np, tg + 2, &ok);
      }
      if (!ok) s_bad = 1;
      __syncthreads();
      cov_finalize(L, D, 1.0 / (double)(n - 1));
      for (int e = t; e < DD; e += kThreads) cov_g[(e / D) * LD + e % D] = L.A[(e / D) * LD + e % D];
      __syncthreads();
      // eigen-free first (the tile stays resident); else the reference's route, which may overlay
      // the tile with the Jacobi buffers
      if (a.fast && spd_fast(L, cov_g, D, want_split, &root_logdet)) root_fast = 1;
      const bool good = root_fast || regularize(L, cov_g, D);
      flag = good ? 1 : 2;
      if (good)
        for (int e = t; e < DD; e += kThreads) st_agent16(b_am + (size_t)e * 2, L.AM[(e / D) * LD + e % D], tg + 3);
      if (t == 0) st_agent16(b_flag, (double)flag, tg + 3);
    }
    // ---- the k-means scale and this part's rows of the scaled copy (bounding.py:1503-1510): beside part 0's solve ----
    if (want_split) {
      if (t < D) L.scale[t] = sqrt(tagged_partner_reduce(b_sq, D, t, np, tg + 2, &ok) / (double)n);
      if (!ok) s_bad = 1;
      __syncthreads();
      double* ps = a.pts_scaled + (size_t)run * a.n * D;
      const int jj = t & (L.DP - 1), p0 = t >> L.DPlog, pstep = kThreads >> L.DPlog;
      if (jj < D) {
        const double sj = L.scale[jj];
        for (int p = s0 + p0; p < s0 + cnt; p += pstep) ps[(size_t)p * D + jj] = v.pts[(size_t)p * D + jj] / sj;
      }
      if (q == 0 && t < D) a.scale_g[(size_t)run * D + t] = L.scale[t];
    }
    if (q > 0) {
      if (t == 0) {
        bool ok1 = true;
        flag = (int)tagged_partner_reduce(b_flag, 1, 0, 1, tg + 3, &ok1);
        if (!ok1) s_bad = 1;
        L.ri[300] = flag;
      }
      __syncthreads();
      flag = L.ri[300];
    }
    if (flag == 1) {
      if (q > 0) {
        for (int e = t; e < DD; e += kThreads) L.AM[(e / D) * LD + e % D] = tagged_partner_reduce(b_am, 1, e, 1, tg + 3, &ok);
        if (!ok) s_bad = 1;
        __syncthreads();
      }
      // ---- Mahalanobis maximum over the own tile ----
      stage_tile(L, v.pts, v.perm, s0, cnt, D, 1);  // still resident unless the Jacobi buffers overlaid it
      double best = -INFINITY;
      if (D >= kMfmaMinDim) {
        best = tile_quadform_max(L, L.AM, cnt, D, best);
      } else {
        for (int p = t; p < cnt; p += kThreads) {
          const double* x = L.tile + p * LD;
          double qf = 0.0;
          for (int i = 0; i < D; ++i) {
            double r = 0.0;
            const double* row = L.AM + i * LD;
            for (int k = 0; k < D; ++k) r = fma(row[k], x[k], r);
            qf = fma(x[i], r, qf);
          }
          best = fmax(best, qf);
        }
      }
      __syncthreads();
      best = block_reduce_max(best, L.red);
      if (t == 0) st_agent16(b_fmx + (size_t)q * 2, best, tg + 4);
      if (q == 0) {
        if (t == 0) {
          bool ok1 = true;
          L.red[0] = tagged_partner_reduce<true>(b_fmx, 1, 0, np, tg + 4, &ok1);
          if (!ok1) s_bad = 1;
        }
        __syncthreads();
        const double fmx = L.red[0];
        __syncthreads();
        if (s_bad) status = DH_ERR_HIP;
        if (root_fast && fmx > 1.0 - kRoundDelta) root_logdet += (double)D * log(fmx / (1.0 - kRoundDelta));
        ellipsoid_rescale(L, cov_g, D, fmx);
        root_fmax = fmin(fmx, 1.0 - kRoundDelta);
        if (status == DH_OK)
          status = root_fast ? ellipsoid_store_fast(L, a, es, cov_g, root_logdet, &lv)
                             : ellipsoid_store(L, a, es, cov_g, &lv);
      }
    } else if (q == 0) {
      // regularised covariance: the reference's second pass (bounding.py:1449-1453) by the
      // single-workgroup routine, from scratch
      __syncthreads();
      if (s_bad) status = DH_ERR_HIP;
      if (status == DH_OK) {
        // the other parts' ranges of the identity permutation were written by other CUs with
        // plain stores (not visible across XCDs inside this kernel): write them here as well
        for (int p = t; p < n; p += kThreads) v.perm[p] = p;
        __threadfence_block();
        __syncthreads();
        status = node_ellipsoid<false>(L, a, v.pts, v.perm, 0, n, es, cov_g, &lv, &root_fmax);
      }
    }
    // ---- status: part 0 alone needs it (it files the root and queues the split); a part that waited in vain says so ----
    __syncthreads();
    if (q > 0) {
      if (s_bad && t == 0) atomicMin(&a.kerr[run], DH_ERR_HIP);  // (folded into the run's status by the next kernel)
    } else {
      if (s_bad) status = DH_ERR_HIP;
      if (t == 0 && status == DH_OK && want_split) queue_split(a, run, 0, 0, n);
    }
    (void)b_stat;
  }
  if (q == 0 && t == 0) {
    Node r;
    r.start = 0;
    r.count = n;
    r.parent = -1;
    r.child0 = r.child1 = -1;
    r.depth = 0;
    r.split = 0;
    r.res_start = 0;
    r.res_len = 0;
    r.fast = root_fast;
    r.has_mean = 0;
    r.logvol = lv;
    r.fmax = root_fmax;
    v.nodes[0] = r;
    a.nnodes_dev[run] = 1;
    a.status[run] = status;
  }
  if (np == 1 && status == DH_OK && a.mode == 0 && n >= 4 * D) {
    {
      // (node_std is a call: handing it a copy keeps L itself out of memory -- with its address taken the carve-up
      // lived on every lane's stack for the whole kernel, 224 bytes written and every L.xxx a scratch load)
      const Lds Lc = L;
      node_std(Lc, v.pts, v.perm, 0, n, D);
      L.c_pts = nullptr;  // the tile now holds what the copy staged
    }
    if (t < D) a.scale_g[(size_t)run * D + t] = L.scale[t];
    double* ps = a.pts_scaled + (size_t)run * a.n * D;
    const int jj = t & (L.DP - 1), p0 = t >> L.DPlog, pstep = kThreads >> L.DPlog;
    if (jj < D) {
      const double sj = L.scale[jj];
      for (int p = p0; p < n; p += pstep) ps[(size_t)p * D + jj] = v.pts[(size_t)p * D + jj] / sj;
    }
    if (t == 0) queue_split(a, run, 0, 0, n);
  }
}

// One part of one splittable node: k-means + partition (node_kmeans_part), and by part 0 the two child records and
// their means.  single = true: the calling workgroup is the node's only part whatever its size -- the node
// must fit the tile (count <= L.TP).
__device__ __forceinline__ void split_body(const RebuildArgs& a, const Lds& L, const RunView& v, int run, int level, int slot, int q,
                           bool single) {
  const int D = a.d, t = threadIdx.x;
  const size_t lp = (size_t)(level & 1) * a.runs + run;
  // k_tree (a.tree): `slot` IS the node; its barrier / done counter / first partial-sum slot sit in nbar
  const bool tq = a.tree == 1;
  int* nb = tq ? a.nbar + ((size_t)run * a.max_nodes + slot) * kBarStride : nullptr;
  const int pb = tq ? ld_agent_i(nb + 2) : (a.tree ? 0 : a.part_base[lp * a.maxw + slot]);
  const int cur = a.tree ? slot : a.split_list[lp * a.maxw + slot];
  const int start = ld_ci(L, &v.nodes[cur].start), count = ld_ci(L, &v.nodes[cur].count),
            depth = ld_ci(L, &v.nodes[cur].depth);
  const int np = single ? 1 : (count + L.TP - 1) / L.TP;
  const int min_size = 2 * D;
  const int KP = 2 * D + 2;
  double* kp0 = a.kpart + ((size_t)run * a.maxp + pb) * KP * 2;  // (value, tag) pairs
  double* kp1 = kp0 + (size_t)a.runs * a.maxp * KP * 2;
  const unsigned long long tag0 = ((unsigned long long)a.epoch << 24) | ((unsigned long long)(level & 0xffff) << 8);
  int* bar = a.tree ? nb : a.kbar + (((size_t)level * a.runs + run) * a.maxw + slot) * kBarStride;
  PH_T0();
  const int n0 = node_kmeans_part(L, a.pts_scaled + (size_t)run * a.n * D, v.perm, v.perm2, start, count, D,
                                  v.estore + (size_t)cur * v.NS, q, np, kp0, kp1, bar, min_size, tag0, level);
  PH_ADD(4);
  LV_T0();
  if (n0 < 0) {
    if (t == 0) atomicMin(a.tree ? &a.status[run] : &a.kerr[run], DH_ERR_HIP);
    return;
  }
  const int n1 = count - n0;
  if (min(n0, n1) < min_size) return;  // reject the split (:1521-1522): node stays a leaf
  // Who creates the children: part 0 in the level pipeline (the next kernel starts after every part has
  // finished); in k_tree the part that finishes LAST -- the children's ellipsoids may be started by other
  // workgroups the moment they are queued, and they read the permutation ranges all parts have just written.
  bool creator = q == 0;
  if (tq) {
    drain_stores();  // this part's permutation range (agent-scope stores) is acknowledged
    __syncthreads();
    if (t == 0)
      L.ri[302] = __hip_atomic_fetch_add(nb + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == np - 1 ? 1 : 0;
    __syncthreads();
    creator = L.ri[302] != 0;
  }
  if (creator && t == 0) {
    L.ri[300] = -1;
    const int c0 = atomicAdd(&a.nnodes_dev[run], 2);
    if (c0 + 2 > a.max_nodes) {
      atomicMin(a.tree ? &a.status[run] : &a.kerr[run], DH_ERR_NOMEM);
    } else {
      Node k0, k1;
      k0.start = start;
      k0.count = n0;
      k1.start = start + n0;
      k1.count = n1;
      k0.parent = k1.parent = cur;
      k0.child0 = k0.child1 = k1.child0 = k1.child1 = -1;
      k0.depth = k1.depth = depth + 1;
      k0.split = k1.split = 0;
      k0.res_start = k1.res_start = 0;
      k0.res_len = k1.res_len = 0;
      k0.fast = k1.fast = 0;
      k0.has_mean = k1.has_mean = 1;
      k0.logvol = k1.logvol = 0.0;
      k0.fmax = k1.fmax = INFINITY;
      if (a.tree) {
        // word by word through agent-scope stores (Node is 16 ints)
        static_assert(sizeof(Node) % 4 == 0, "Node is copied as ints");
        const int* w0 = (const int*)&k0;
        const int* w1 = (const int*)&k1;
        int* d0 = (int*)&v.nodes[c0];
        int* d1 = (int*)&v.nodes[c0 + 1];
        for (int i = 0; i < (int)(sizeof(Node) / 4); ++i) {
          st_agent_i(d0 + i, w0[i]);
          st_agent_i(d1 + i, w1[i]);
        }
        st_agent_i(&v.nodes[cur].child0, c0);
        st_agent_i(&v.nodes[cur].child1, c0 + 1);
        st_agent_i(&v.nodes[cur].split, 1);
      } else {
        v.nodes[c0] = k0;
        v.nodes[c0 + 1] = k1;
        v.nodes[cur].child0 = c0;
        v.nodes[cur].child1 = c0 + 1;
        v.nodes[cur].split = 1;
        const int e = atomicAdd(&a.nell[(size_t)level * a.runs + run], 2);
        int* el = a.ell_list + ((size_t)level * a.runs + run) * 2 * a.maxw;
        el[e] = c0;
        el[e + 1] = c0 + 1;
      }
      L.ri[300] = c0;
    }
  }
  // The children's means come for free: the centroids after the tenth update ARE the means of the final
  // clusters (cluster sums / counts, in the scaled coordinates of the k-means).  k_ell then skips its
  // mean pass over the points (one of its three gathers).
  if (creator) {
    if (t == 0 && !(L.ri[300] >= 0 && L.ri[300] + 2 <= a.max_nodes)) L.ri[300] = -1;
    __syncthreads();
    const int c0 = L.ri[300];
    if (c0 >= 0 && t < 2 * D) {
      const int c = t >= D ? 1 : 0, j = t - c * D;
      st_c(L, v.estore + (size_t)(c0 + c) * v.NS + j, L.cen[c * D + j] * L.scale[j]);
    }
    if (a.tree && c0 >= 0) {
      drain_stores();  // tree entries and means acknowledged before the children are published
      __syncthreads();
      if (tq && t == 0) (void)tq_push(a, false, run, c0, 2);
    }
  }
  LV_ADD(level, 3);
}

// gp = the parts per run this launch provides (the level's bound, <= a.maxp).  Run-minor like k_ell -- the q-th part
// of every run before anyone's (q + 1)-th -- but only inside chunks of cr runs whose cr * gp workgroups can all be
// resident at once: the parts of a node meet at spin barriers, workgroups are dispatched in index order, and a chunk
// that fits the chip can always be completed by the workgroups in front of it finishing (chunks are dispatched one
// after the other; cr = 1 is the run-major order of rounds 2-4).
__device__ __forceinline__ void k_split_impl(const RebuildArgs& a, int level, int gp, int cr);
#ifndef DH_KSPLIT_WAVES
#define DH_KSPLIT_WAVES 5  // waves per SIMD the register allocation leaves room for: five 31 KB parts share a CU
#endif
__global__ void __launch_bounds__(kThreads, DH_KSPLIT_WAVES) k_split(RebuildArgs a, int level, int gp, int cr) {
  WG_STAMP(0, level, 0);
  PH_LEVEL(-1);
  k_split_impl(a, level, gp, cr);
  WG_STAMP(0, level, 1);
}
__device__ __forceinline__ void k_split_impl(const RebuildArgs& a, int level, int gp, int cr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int per = cr * gp, chunk = blockIdx.x / per, b = blockIdx.x - chunk * per;
  const int rc = min(cr, a.runs - chunk * cr);  // runs of this chunk (the last one may be short)
  const int run = chunk * cr + b % rc, ps = b / rc;
  if (ps >= gp) return;
  const int nps = a.nparts[(size_t)level * a.runs + run];
  if (nps > gp) {  // (cannot happen: a level's parts are bounded by n / tps + its node count)
    if (threadIdx.x == 0) atomicMin(&a.kerr[run], DH_ERR_NOMEM);
    return;
  }
  if (ps >= nps) return;
  // status is only written by the other kernels of the pipeline (errors of THIS kernel go
  // to kerr): all parts of a node take the same decision here
  if (a.status[run] != DH_OK) return;
  const int D = a.d, t = threadIdx.x;
  Lds L;
  carve_split(L, smem, D, a.tps);
  const RunView v = view_of(a, run, L.LD);
  if (t < D) L.scale[t] = a.scale_g[(size_t)run * D + t];
  __syncthreads();
  const size_t lp = (size_t)(level & 1) * a.runs + run;
  const int slot = a.part_list[(lp * a.maxp + ps) * 2], q = a.part_list[(lp * a.maxp + ps) * 2 + 1];
  split_body(a, L, v, run, level, slot, q, false);
}

// The bounding ellipsoid of one new child and, if it is big enough, its entry in the next level's split list.
// Returns false after an error (status set).
// DEFER (the level kernel's eigen-free form): a node the eigen-free path declines -- or one without a mean in its
// record, which a child never is -- goes to the work-queue tail untouched (its record is written last, so nothing of
// it exists yet), as k_ell_wave's declined leaves do.
template <bool SLOW, bool DEFER = false>
__device__ __forceinline__ bool ell_body(const RebuildArgs& a, const Lds& L, const RunView& v, int run, int level, int node) {
  const int D = a.d, t = threadIdx.x;
  const int start = ld_ci(L, &v.nodes[node].start), count = ld_ci(L, &v.nodes[node].count);
  double lv = 0.0, fmx = INFINITY;
  __syncthreads();
  L.c_pts = nullptr;
  int rc;
  if constexpr (DEFER) {
    rc = v.nodes[node].has_mean ? node_ellipsoid<true, true>(L, a, v.pts, v.perm, start, count, v.estore + (size_t)node * v.NS,
                                                             v.estore + (size_t)node * v.NS + v.ES, &lv, &fmx, true)
                                : kDeferred;
    if (rc == kDeferred) {
      if (t == 0) (void)tq_push(a, false, run, node, 1);
      return true;
    }
  } else {
    rc = node_ellipsoid<!SLOW>(L, a, v.pts, v.perm, start, count, v.estore + (size_t)node * v.NS,
                               v.estore + (size_t)node * v.NS + v.ES, &lv, &fmx,
                               ld_ci(L, &v.nodes[node].has_mean) != 0);
  }
  const bool full = SLOW || rc == kFullRecord;
  if (rc != DH_OK && rc != kFullRecord) {
    set_status(a, run, rc);
    return false;
  }
  if (a.tree) {  // the record (agent-scope stores of all threads) is acknowledged before the split is queued
    drain_stores();
    __syncthreads();
  }
  if (t == 0) {
    v.nodes[node].logvol = lv;
    v.nodes[node].fmax = fmx;
    v.nodes[node].fast = full ? 0 : 1;
    if (count >= 4 * D) {  // big enough to try a split at the next level (:1492-1496)
      if (!a.tree && a.tree_from > a.levels && level + 1 >= a.levels) {
        atomicMin(&a.status[run], DH_ERR_NOMEM);  // deeper than the launch plan
      } else {
        queue_split(a, run, level + 1, node, count);
      }
    }
  }
  return true;
}

// One workgroup per new child.  SLOW = false: the eigen-free path (with its in-place fallback);
// SLOW = true: the reference's route for every node -- the kernel of the diagnostic mode
// DH_REBUILD_FAST=0.  Two kernels so that the common one stays small (registers: two workgroups per CU).
// DEFER (round 6): the eigen-free form alone -- what it declines goes to the work-queue tail (launched whenever there is one)
template <bool SLOW, bool DEFER>
// (two workgroups per CU: held to the 168 registers of three, with a 128-point tile so that LDS would allow it, the
// eigen-free path spills and the rebuild loses 7 %)
__device__ __forceinline__ void k_ell_impl(const RebuildArgs& a, int level, int G, int skip_done, int leaf_cap, int tp);
template <bool SLOW, bool DEFER = false>
__global__ void __launch_bounds__(kThreads, SLOW ? 1 : 2) k_ell(RebuildArgs a, int level, int G, int skip_done, int leaf_cap,
                                                                 int tp) {
  WG_STAMP(1, level, 0);
  PH_LEVEL(level);
  k_ell_impl<SLOW, DEFER>(a, level, G, skip_done, leaf_cap, tp);
  WG_STAMP(1, level, 1);
}
template <bool SLOW, bool DEFER>
__device__ __forceinline__ void k_ell_impl(const RebuildArgs& a, int level, int G, int skip_done, int leaf_cap, int tp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // run-minor: the g-th node of EVERY run before anyone's (g + 1)-th -- workgroups are dispatched in index order at
  // a finite rate (~30 per us), and with the runs major the last run's first node started after 2 600 others
  const int run = blockIdx.x % a.runs, g = blockIdx.x / a.runs;
  const int* list = a.ell_list + ((size_t)level * a.runs + run) * 2 * a.maxw;
  const int cnt = a.nell[(size_t)level * a.runs + run];
  if (g >= cnt) return;
  if (a.kerr[run] != DH_OK) {  // raised by a k_split workgroup of this level
    if (threadIdx.x == 0) atomicMin(&a.status[run], a.kerr[run]);
    return;
  }
  if (a.status[run] != DH_OK) return;
  const int D = a.d;
  Lds L;
  carve(L, smem, D, tp);  // tp points staged at once: 256, or 512 at the top levels (the launcher's choice)
  const RunView v = view_of(a, run, L.LD);
  for (int slot = g; slot < cnt; slot += G) {
    // (a child is created with fmax = inf: a finite value = k_ell_wave has built this one)
    if (skip_done && v.nodes[list[slot]].fmax < INFINITY) continue;
    if (leaf_cap > 0) {  // the leaves of this level are k_ell_wave<128>'s, on the side stream (the same test as its own)
      const Node& nd = v.nodes[list[slot]];
      if (nd.has_mean && nd.count >= 2 && nd.count <= leaf_cap && nd.count < 4 * D) continue;
    }
    if (!ell_body<SLOW, DEFER>(a, L, v, run, level, list[slot])) return;
  }
}

// ---- small nodes, one wavefront each ---------------------------------------------------------------------
// The deep levels of a tree are many small nodes (the 64-run bench rebuild: 2 048 leaves of ~62 points at level
// 5; an eggbox live set: hundreds of nodes of 8-60 points per level and run), and a level's time is the number of
// ROUNDS its nodes need on the chip's workgroup slots: k_ell holds 256 threads, 242 registers and 77 KB of LDS per
// node -- two nodes per CU.  k_ell_wave builds a node of at most `cap` points with ONE wavefront and an LDS carve of
// its own (tile of cap points, two or four D x D matrices): the same routines instantiated for 64 threads
// (stage_tile / spd_fast / node_fmax / ... <64>: every element-wise loop and every MFMA tile is the same
// instruction on the same operands whichever thread issues it), the covariance contraction with the four waves of
// k_ell played in turn (tile_cov_accumulate's wsel: wave w takes points [64 w, 64 w + 64), partial sums folded in
// wave order).  So a node gets the SAME BITS from either kernel (tests/test_gpu_edges.py holds the whole tree to
// that), and which kernel builds it is a scheduling decision: k_ell_wave runs first over the level's list and takes
// the nodes that fit (`axis` = 0: leaves only -- count < 4 D, no major axis wanted, half the matrices), marks them
// by their finite fmax, and k_ell skips those.  A node whose eigen-free path does not apply (spd_fast false) is
// left untouched for k_ell's in-place fallback.
constexpr int kWaveDeclined = 2;

__host__ __device__ inline size_t wave_lds_bytes(int D, int cap, bool axis) {
  const int LD = D | 1;
  return ((((size_t)cap * LD + (axis ? 4 : 2) * (size_t)D * LD + 2 * (size_t)D + 64) * 8) + 15) & ~(size_t)15;
}

__device__ __forceinline__ void carve_wave(Lds& L, unsigned char* smem, int D, int cap, bool axis) {
  L.LD = D | 1;
  L.TP = cap;
  L.c_pts = nullptr;
  L.c_start = L.c_cnt = L.c_how = -1;
  L.coh = false;
  L.KG = 1;
  L.DP = 1;
  L.DPlog = 0;
  while (L.DP < D) {
    L.DP <<= 1;
    ++L.DPlog;
  }
  double* p = (double*)smem;
  L.tile = p;
  p += (size_t)cap * L.LD;
  L.A = p;
  p += D * L.LD;
  L.AM = p;
  p += D * L.LD;
  if (axis) {
    L.V = p;
    p += D * L.LD;
    L.AX = p;
    p += D * L.LD;
  } else {
    L.V = nullptr;  // (the squarings of the major axis: not wanted for a leaf)
    L.AX = L.A;     // spd_fast zeroes AX after its last use of A
  }
  L.mean = p;
  p += D;
  L.lam = p;
  p += D;
  L.red = p;
  L.ibuf = (int*)L.red;
  L.ibuf_cap = 128;  // (64 doubles)
  L.scale = L.cen = L.sums = L.rc = L.rs = L.kred = nullptr;
  L.ri = L.perm_sort = nullptr;
  L.JA[0] = L.JA[1] = L.JV[0] = L.JV[1] = nullptr;
  L.JLD = 0;
  L.j_alias = false;
}

// node_ellipsoid<true> for a child (mean in its record) of at most L.TP points, by NT / 64 wavefronts (one, or two:
// round 5).  With two, wavefront w takes k_ell's wavefronts w, w + 2 in turn and the partial sums are still folded in
// k_ell's wave order: the same bits from every form.
template <int NT>
__device__ __forceinline__ int node_ellipsoid_wave(const Lds& L, const RebuildArgs& a, const double* pts, const int* perm,
                                                   int start, int count, double* es, double* cov_g,
                                                   double* logvol_out, double* fmax_out) {
  constexpr int NW = NT / 64;
  const int D = a.d, t = threadIdx.x, LD = L.LD;
  if (t < D) L.mean[t] = es[t];
  __syncthreads();
  // node_cov, one tile
  stage_tile<NT>(L, pts, perm, start, count, D, 1);
  {
    const int nb = (D + 15) >> 4, lane = t & 63, lj = lane & 15, lk = lane >> 4, mw = t >> 6;
    for (int wv0 = 0; wv0 * 64 < count; wv0 += NW) {
      const int wv = wv0 + mw;  // the wave of k_ell this wavefront plays now
      const bool mine = wv * 64 < count;
      mfma_acc acc[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[b] = (mfma_acc){0.0, 0.0, 0.0, 0.0};
      if (mine) tile_cov_accumulate(L, count, D, acc, wv);
      for (int turn = 0; turn < NW; ++turn) {  // cov_fold_waves, one wave's turn after the other
        if (mine && mw == turn) {
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            const int ib = b == 0 ? 0 : b == 1 ? 0 : b == 2 ? 1 : b == 3 ? 0 : b == 4 ? 1 : 2;
            const int jb = b == 0 ? 0 : b == 1 ? 1 : b == 2 ? 1 : 2;
            if (jb < nb) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = ib * 16 + lk + 4 * r, j = jb * 16 + lj;
                if (i < D && j < D) {
                  double v = acc[b][r];
                  if (wv > 0) v += L.A[i * LD + j];
                  L.A[i * LD + j] = v;
                }
              }
            }
          }
        }
        __syncthreads();
      }
    }
  }
  cov_finalize<NT>(L, D, 1.0 / (double)(count - 1));
  for (int e = t; e < D * D; e += NT) cov_g[(e / D) * LD + e % D] = L.A[(e / D) * LD + e % D];
  __syncthreads();
  double logdet = 0.0;
  if (!spd_fast<NT>(L, cov_g, D, a.mode == 0 && count >= 4 * D, &logdet)) return kWaveDeclined;
  const double fmx = node_fmax<NT>(L, pts, perm, start, count, D);
  if (fmx > 1.0 - kRoundDelta) logdet += (double)D * log(fmx / (1.0 - kRoundDelta));
  ellipsoid_rescale<NT>(L, cov_g, D, fmx);
  *fmax_out = fmin(fmx, 1.0 - kRoundDelta);
  return ellipsoid_store_fast<NT>(L, a, es, cov_g, logdet, logvol_out);
}

// NT = 128 (round 5): the leaves of a level at D >= 14 by two wavefronts each on the side stream, beside the level
// kernels of the splittable nodes (a leaf is only read by k_finish); a node this kernel declines is queued for the
// work-queue tail (`defer`), which runs after the side stream has joined.
template <int NT>
__global__ void __launch_bounds__(NT, 3) k_ell_wave(RebuildArgs a, int level, int G, int cap, int axis, int defer) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  const int run = blockIdx.x % a.runs, g = blockIdx.x / a.runs;  // (run-minor, as k_ell)
  const int* list = a.ell_list + ((size_t)level * a.runs + run) * 2 * a.maxw;
  const int cnt = a.nell[(size_t)level * a.runs + run];
  if (g >= cnt) return;
  if (a.kerr[run] != DH_OK || a.status[run] != DH_OK) return;  // (k_ell, next on the stream, reports it)
  const int D = a.d, t = threadIdx.x;
  Lds L;
  carve_wave(L, smem, D, cap, axis != 0);
  const RunView v = view_of(a, run, L.LD);
  for (int slot = g; slot < cnt; slot += G) {
    const int node = list[slot];
    const int start = v.nodes[node].start, count = v.nodes[node].count;
    if (!v.nodes[node].has_mean || count < 2 || count > cap || (!axis && count >= 4 * D)) continue;
    double lv = 0.0, fmx = INFINITY;
    __syncthreads();
    L.c_pts = nullptr;
    const int rc = node_ellipsoid_wave<NT>(L, a, v.pts, v.perm, start, count, v.estore + (size_t)node * v.NS,
                                           v.estore + (size_t)node * v.NS + v.ES, &lv, &fmx);
    if (rc == kWaveDeclined) {
      // (the node is untouched: fmax = inf, no record)
      if (defer && t == 0) (void)tq_push(a, false, run, node, 1);
      continue;
    }
    if (rc != DH_OK) {
      set_status(a, run, rc);
      return;
    }
    if (t == 0) {  // ell_body's tail
      v.nodes[node].logvol = lv;
      v.nodes[node].fast = 1;
      if (count >= 4 * D) {
        if (a.tree_from > a.levels && level + 1 >= a.levels)
          atomicMin(&a.status[run], DH_ERR_NOMEM);
        else
          queue_split(a, run, level + 1, node, count);
      }
      v.nodes[node].fmax = fmx;
    }
  }
}

// the two item routines are calls, not inlined code: inlined into one loop body their register needs add
// up past the 256 of two workgroups per CU (42 VGPRs spilled); behind a call each has its own allocation and
// the loop keeps nothing live across it but the item
__device__ __attribute__((noinline)) void tree_split_item(const RebuildArgs& a, unsigned char* smem, int run, int node, int part) {
  const int D = a.d, t = threadIdx.x;
  Lds LS;
  carve_split(LS, smem, D, a.tps);
  LS.coh = !a.tq_nocoh;
  const RunView v = view_of(a, run, LS.LD);
  if (t < D) LS.scale[t] = a.scale_g[(size_t)run * D + t];
  __syncthreads();
  split_body(a, LS, v, run, 0, node, part, false);
}
__device__ __attribute__((noinline)) void tree_ell_item(const RebuildArgs& a, unsigned char* smem, int run, int node) {
  Lds L;
  carve(L, smem, a.d);
  L.coh = !a.tq_nocoh;
  const RunView v = view_of(a, run, L.LD);
  (void)ell_body<false>(a, L, v, run, 0, node);
}

// (the item routines take the argument block by reference, so the worker keeps a copy of it on its stack: 476 bytes
// per lane.  Behind a call the copy is made only by workers that have work: in the kernel's own prologue it was 58 MB
// of scratch stores -- the whole 10 us of the empty-queue launch that every rebuild ends with.)
__device__ __attribute__((noinline)) void tree_worker(RebuildArgs a, unsigned char* smem) {
  __shared__ unsigned long long s_item;
  const int t = threadIdx.x;
  for (;;) {
    __syncthreads();  // the previous item's LDS use is over
    if (t == 0) {
      unsigned long long it = 0;
      const int ticket = __hip_atomic_fetch_add(a.tq_ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket < a.tq_cap) {
        long long spins = 0;
        for (;;) {
          it = __hip_atomic_load(a.tq_items + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (it) break;
          if ((spins & 3) == 0 && __hip_atomic_load(a.tq_ctl + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) break;
          for (int k = 0; k < a.tq_sleep; ++k) __builtin_amdgcn_s_sleep(8);
          if (++spins > (1ll << 22)) {  // a producer died: fail loudly instead of hanging the device
            __hip_atomic_store(a.tq_ctl + 48, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      s_item = it;
    }
    __syncthreads();
    const unsigned long long it = s_item;
    if (!it) return;
    const int run = (int)((it >> 40) & 0x3fffff), node = (int)((it >> 16) & 0xffffff), part = (int)(it & 0xffff);
    if (it & kItemSplit)
      tree_split_item(a, smem, run, node, part);
    else
      tree_ell_item(a, smem, run, node);
    drain_stores();
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add(a.tq_ctl + 32, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(kThreads, 2) k_tree(RebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  // nothing queued by the level kernels (they are complete: stream order) = nothing ever will be
  if (__hip_atomic_load(a.tq_ctl + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) return;
  tree_worker(a, smem);
}

// ---- small subtrees, one workgroup each (round 4) -----------------------------------------------------------
// From the level where a child has a few hundred points the level pipeline is bound by workgroup slots and by its
// kernel boundaries, not by work: the 64-run bench tree has 512 / 1 024 / 2 048 children of ~250 / 125 / 62 points at
// levels 3-5, every level a k_split + k_ell pair of ~225 us in which a node waits for the slowest of its level twice.
// A child that fits the 256-point tile cannot have a big descendant, so ONE workgroup takes its whole subtree --
// ellipsoid, k-means of the whole node (no parts, no barrier between workgroups), the children's ellipsoids, ... --
// depth first over a small stack, with the node routines of the level kernels (same arithmetic; the tile's
// wavefronts are grouped like the 128-point parts, so a node's sums come out as the level kernels' would) and, as
__global__ void __launch_bounds__(kThreads) k_finish(RebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  const int D = a.d, t = threadIdx.x, run = blockIdx.x;
  const int n = a.n_arr ? a.n_arr[run] : a.n;
  if (a.active && !a.active[run]) return;
  Lds L;
  carve(L, smem, D);
  const RunView v = view_of(a, run, L.LD);
  Node* nodes = v.nodes;
  int* reslist = v.reslist;
  const double* estore = v.estore;
  const int NS = v.NS, DD = D * D;
  int status = min(a.status[run], a.kerr[run]);  // k_split errors of the last level are folded here
  if (a.tq_ctl && a.tq_ctl[48]) status = min(status, (int)DH_ERR_HIP);  // a k_tree worker gave up waiting
  const int nnodes = min(a.nnodes_dev[run], a.max_nodes);
  // the accept test below is a serial walk over the tree by one thread: every access to a
  // node in global memory is a dependent ~1 us load, so the tree (and, when it fits, the
  // result-list arena) is copied to LDS first
  if (a.fin_extra_off) {
    Node* nl = (Node*)(smem + a.fin_extra_off);
    for (int i = t; i < nnodes; i += kThreads) nl[i] = nodes[i];
    nodes = nl;
    if (a.fin_res_lds) reslist = (int*)(nl + a.max_nodes);
  }
  __syncthreads();

  // ---- bottom-up accept test (bounding.py:1541-1563), level-parallel ----
  // A node's verdict needs only its children's (list length, logsumexp of the list's volumes),
  // so the tree is reduced depth by depth with all threads (the eggbox-like C3 tree has 1 719
  // nodes: a serial walk cost 1.3 ms); a top-down pass then hands every surviving leaf its
  // position in the output list (child-0 subtree first: the reference's concatenation order).
  if (status == DH_OK) {
    double* lse = a.fin_lse + (size_t)run * a.max_nodes;
    int* acc = a.fin_int + (size_t)run * a.max_nodes * 2;  // accepted split?
    int* act = acc + a.max_nodes;                           // on the output path?
    int md = 0;
    for (int i = t; i < nnodes; i += kThreads) act[i] = 0;
    __syncthreads();
    {
      int mine = 0;
      for (int i = t; i < nnodes; i += kThreads) mine = max(mine, nodes[i].depth);
      mine = max(mine, xor_lane_i32<32>(mine));
      mine = max(mine, xor_lane_i32<16>(mine));
      mine = max(mine, xor_lane_i32<8>(mine));
      mine = max(mine, xor_lane_i32<4>(mine));
      mine = max(mine, xor_lane_i32<2>(mine));
      mine = max(mine, xor_lane_i32<1>(mine));
      if ((t & 63) == 0) L.ri[t >> 6] = mine;
      __syncthreads();
      md = max(max(L.ri[0], L.ri[1]), max(L.ri[2], L.ri[3]));
      __syncthreads();
    }
    const int nparam = (D * (D + 3)) / 2;
    for (int dpt = md; dpt >= 0; --dpt) {
      for (int i = t; i < nnodes; i += kThreads) {
        const Node nd = nodes[i];
        if (nd.depth != dpt) continue;
        int rl = 1, ac = 0;
        double ls = nd.logvol;
        if (nd.split) {
          const Node& k0 = nodes[nd.child0];
          const Node& k1 = nodes[nd.child1];
          const int len = k0.res_len + k1.res_len;
          const double dec = nparam * log((double)nd.count) / (double)nd.count;
          const double both = logaddexp_d(lse[nd.child0], lse[nd.child1]);
          const bool accept = ((logaddexp_d(k0.logvol, k1.logvol) - nd.logvol) < -dec) ||
                              ((both - nd.logvol) < -dec * (len - 1));
          if (accept) {
            rl = len;
            ls = both;
            ac = 1;
          }
        }
        nodes[i].res_len = rl;
        lse[i] = ls;
        acc[i] = ac;
      }
      __threadfence_block();
      __syncthreads();
    }
    if (t == 0) {
      nodes[0].res_start = 0;
      act[0] = 1;
    }
    __threadfence_block();
    __syncthreads();
    for (int dpt = 0; dpt <= md; ++dpt) {
      for (int i = t; i < nnodes; i += kThreads) {
        const Node nd = nodes[i];
        if (nd.depth != dpt || !act[i]) continue;
        if (acc[i]) {
          nodes[nd.child0].res_start = nd.res_start;
          nodes[nd.child1].res_start = nd.res_start + nodes[nd.child0].res_len;
          act[nd.child0] = 1;
          act[nd.child1] = 1;
        } else if (nd.res_start < a.reslist_cap) {
          reslist[nd.res_start] = i;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
  }

  // ---- emit the ellipsoid list ----
  int M = 0;
  if (status == DH_OK) {
    M = nodes[0].res_len;
    if (M > a.max_ells) status = DH_ERR_NOMEM;
  }
  if (status == DH_OK) {
    const int rs0 = 0;
    double* o_ctr = a.ctrs + (size_t)run * a.max_ells * D;
    double* o_cov = a.covs + (size_t)run * a.max_ells * DD;
    double* o_am = a.ams + (size_t)run * a.max_ells * DD;
    double* o_ax = a.axes + (size_t)run * a.max_ells * DD;
    double* o_al = a.axlens + (size_t)run * a.max_ells * D;
    double* o_lv = a.logvols + (size_t)run * a.max_ells;
    // (round 6) the records of ALL M ellipsoids in flight together: one ellipsoid after the other, each with its own
    // cold reads, the copy of a 15-ellipsoid list (eggbox) was 15 round trips in a row -- half of this kernel at C3
    for (int x = t; x < M * D; x += kThreads) {
      const int m = x / D, e = x - m * D;
      const double* es = estore + (size_t)reslist[rs0 + m] * NS;
      o_ctr[x] = es[e];
      o_al[x] = es[D + 3 * DD + e];
    }
    // (every load of a thread first, then its stores: the compiler cannot move a load of the record above a store to
    // an output it may alias, and a serial chain of cold round trips was most of this phase)
    for (int x0 = t; x0 < M * DD; x0 += 4 * kThreads) {
      double c[4], p[4], xx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = x0 + u * kThreads, xc = x < M * DD ? x : 0;
        const int m = xc / DD, ec = xc - m * DD;
        const double* es = estore + (size_t)reslist[rs0 + m] * NS;
        c[u] = es[D + ec];
        p[u] = es[D + DD + ec];
        xx[u] = es[D + 2 * DD + ec];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = x0 + u * kThreads;
        if (x < M * DD) {
          o_cov[x] = c[u];
          o_am[x] = p[u];
          o_ax[x] = xx[u];
        }
      }
    }
    __syncthreads();  // (the root's side-stream eigen-system below overwrites what other threads have just copied)
    for (int m = 0; m < M; ++m) {
      const int ni = reslist[rs0 + m];
      int need_eig = nodes[ni].fast;
      if (need_eig && ni == 0 && a.root_eig) {
        // the root's eigen-system was solved on the side stream (k_root_eig)
        const double* re = a.root_eig + (size_t)run * (2 * DD + D + 2);
        if (re[2 * DD + D + 1] == 1.0) {
          for (int e = t; e < DD; e += kThreads) {
            o_am[(size_t)m * DD + e] = re[e];
            o_ax[(size_t)m * DD + e] = re[DD + e];
          }
          for (int e = t; e < D; e += kThreads) o_al[m * D + e] = re[2 * DD + e];
          if (t == 0) o_lv[m] = re[2 * DD + D];
          need_eig = 0;
        }
      } else if (t == 0) {
        o_lv[m] = nodes[ni].logvol;
      }
      if (need_eig && t == 0) o_lv[m] = nodes[ni].logvol;
      if (t == 0 && a.out_node) {
        a.out_node[(size_t)run * a.max_ells + m] = ni;
        a.out_fast[(size_t)run * a.max_ells + m] = need_eig;
      }
      if (a.leaf_of_point) {
        int* lop = a.leaf_of_point + (size_t)run * a.n;
        const int s0 = nodes[ni].start, c = nodes[ni].count;
        for (int p = t; p < c; p += kThreads) lop[v.perm[s0 + p]] = m;
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- coverage check: every point inside some ellipsoid, strict < 1
    //      (MultiEllipsoid.update, bounding.py:683-685).  mode 1 has no check.
    if (a.mode == 0) {
      // fast path: the leaves partition the points, so "every point lies in its own leaf's
      // ellipsoid" (max of the quadratic form over the leaf < 1, the matrix-core pass of
      // node_fmax) already proves coverage with n instead of n x M quadratic forms
      // ... and that maximum was taken when the leaf's ellipsoid was built and rescaled (Node.fmax:
      // 1 - 1e-3 after the rescale, or the measured value): a list whose leaves all carry a value
      // safely below 1 is covered without touching the points again
      double worst = -INFINITY;
      bool known = true;
      for (int m = 0; m < M; ++m) {
        const double f = nodes[reslist[rs0 + m]].fmax;
        if (!(f <= 1.0 - 0.5 * kRoundDelta)) known = false;
        worst = fmax(worst, f);
      }
      if (!known) worst = -INFINITY;
      for (int m = 0; m < M && !known; ++m) {
        const int ni = reslist[rs0 + m];
        __syncthreads();
        for (int e = t; e < DD; e += kThreads) L.AM[(e / D) * L.LD + e % D] = o_am[(size_t)m * DD + e];
        if (t < D) L.mean[t] = o_ctr[m * D + t];
        __syncthreads();
        L.c_pts = nullptr;  // L.mean changed: a cached centred tile is stale
        worst = fmax(worst, node_fmax(L, v.pts, v.perm, nodes[ni].start, nodes[ni].count, D));
      }
      if (!(worst < 1.0)) {
        // some point is outside its own leaf: the reference's full test (any ellipsoid)
        int uncovered = 0;
        for (int base = 0; base < n; base += L.TP) {
          const int cnt = min(L.TP, n - base);
          stage_tile(L, v.pts, v.perm, base, cnt, D, 0);
          bool inside = false;
          for (int m = 0; m < M; ++m) {
            for (int e = t; e < DD; e += kThreads) L.AM[(e / D) * L.LD + e % D] = o_am[(size_t)m * DD + e];
            if (t < D) L.mean[t] = o_ctr[m * D + t];
            __syncthreads();
            if (t < cnt && !inside) {
              const double* x = L.tile + t * L.LD;
              double q = 0.0;
              for (int i = 0; i < D; ++i) {
                double r = 0.0;
                const double* row = L.AM + i * L.LD;
                for (int jj = 0; jj < D; ++jj) r = fma(row[jj], x[jj] - L.mean[jj], r);
                q = fma(x[i] - L.mean[i], r, q);
              }
              if (q < 1.0) inside = true;
            }
            __syncthreads();
          }
          if (t < cnt && !inside) uncovered = 1;
        }
        const int tot = block_reduce_sum_int(uncovered, L.ri);
        if (tot > 0) status = DH_ERR_REGION;
      }
    }
  }
  if (t == 0) {
    a.nells[run] = (status == DH_OK) ? M : 0;
    a.status[run] = status;
    if (a.nnodes_out) a.nnodes_out[run] = nnodes;
  }
}

// ---- the root's eigen-system, speculatively -------------------------------------------------------
// For a unimodal live set the accept test keeps the root alone, and its eigh would sit on the
// critical path behind k_finish.  It needs nothing but the root's final covariance, which k_root
// leaves behind: this kernel runs on the context's side stream next to the level kernels and files
// the result (am | axes | axlens | logvol | ok) for k_finish to pick up if the root is an output.
__global__ void __launch_bounds__(kThreads) k_root_eig(RebuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  const int run = blockIdx.x, D = a.d, DD = D * D, t = threadIdx.x;
  double* re = a.root_eig + (size_t)run * (2 * DD + D + 2);
  if (a.active && !a.active[run]) return;
  if (t == 0) re[2 * DD + D + 1] = 0.0;
  if (a.status[run] != DH_OK) return;
  Lds L;
  carve(L, smem, D);
  const RunView v = view_of(a, run, L.LD);
  const int LD = L.LD;
  if (!v.nodes[0].fast) return;       // the reference route already left the full record
  double* cov_g = v.estore + v.ES;    // node 0: the final (rescaled) covariance; a "good" matrix is not modified
  (void)regularize(L, cov_g, D);
  bool ok = true;
  double slog = 0.0;
  for (int k = 0; k < D; ++k) {
    const double l = L.lam[k];
    if (!(l > 0.0) || !isfinite(l)) ok = false;
    slog += log(l);
  }
  for (int e = t; e < DD; e += kThreads) {
    const int i = e / D, j = e - i * D;
    re[e] = L.AM[i * LD + j];
    re[DD + e] = L.AX[i * LD + j];
  }
  if (t < D) re[2 * DD + t] = sqrt(L.lam[t]);
  __syncthreads();
  if (t == 0) {
    re[2 * DD + D] = a.prefactor + 0.5 * slog;
    re[2 * DD + D + 1] = ok ? 1.0 : 0.0;
  }
}

// ---- the eigen-system of the OUTPUT ellipsoids -----------------------------------------------------
// Tree nodes take the eigen-free path (spd_fast); the few that survive the accept test get what
// Ellipsoid.__init__ computes (bounding.py:201-240: eigh of the final covariance -> axes, axis
// lengths, log-volume; am = V diag(1/lam) V^T as improve_covar_mat forms it) here, one workgroup per
// output ellipsoid, all runs at once.  grid = runs x G; workgroup g takes outputs g, g + G, ...
__global__ void __launch_bounds__(kThreads) k_out_eig(RebuildArgs a, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  const int run = blockIdx.x / G, g = blockIdx.x % G;
  if (a.active && !a.active[run]) return;
  if (a.status[run] != DH_OK) return;
  const int M = a.nells[run];
  if (g >= M) return;
  const int D = a.d, DD = D * D, t = threadIdx.x;
  Lds L;
  carve(L, smem, D);
  const RunView v = view_of(a, run, L.LD);
  const int LD = L.LD;
  for (int m = g; m < M; m += G) {
    if (!a.out_fast[(size_t)run * a.max_ells + m]) continue;
    const int ni = a.out_node[(size_t)run * a.max_ells + m];
    double* cov_g = v.estore + (size_t)ni * v.NS + v.ES;  // the node's final (rescaled) covariance, D x LD
    __syncthreads();
    (void)regularize(L, cov_g, D);  // the certificate of spd_fast says "good": a plain eigh
    bool ok = true;
    double slog = 0.0;
    for (int k = 0; k < D; ++k) {
      const double l = L.lam[k];
      if (!(l > 0.0) || !isfinite(l)) ok = false;
      slog += log(l);
    }
    if (!ok) {
      set_status(a, run, DH_ERR_VALUE);
      if (t == 0) a.nells[run] = 0;
      return;
    }
    double* o_cov = a.covs + ((size_t)run * a.max_ells + m) * DD;
    double* o_am = a.ams + ((size_t)run * a.max_ells + m) * DD;
    double* o_ax = a.axes + ((size_t)run * a.max_ells + m) * DD;
    double* o_al = a.axlens + ((size_t)run * a.max_ells + m) * D;
    for (int e = t; e < DD; e += kThreads) {
      const int i = e / D, j = e - i * D;
      o_cov[e] = cov_g[i * LD + j];
      o_am[e] = L.AM[i * LD + j];
      o_ax[e] = L.AX[i * LD + j];
    }
    if (t < D) o_al[t] = sqrt(L.lam[t]);
    if (t == 0) a.logvols[(size_t)run * a.max_ells + m] = a.prefactor + 0.5 * slog;
  }
}

// ---------------------------------------------------------------------------
// improve_covar_mat (bounding.py:1311-1384) as a call of its own: one workgroup per matrix runs
// the very `regularize` routine of the rebuild pipeline on a caller-supplied matrix.
// work: m x D x LD (odd leading dimension), in: the matrix, out: the returned covariance.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
    improve_cov_kernel(int m, int D, double* work, int* good, double* covs, double* ams, double* axes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PH_LEVEL(-1);
  Lds L;
  carve(L, smem, D);
  const int e = blockIdx.x, t = threadIdx.x, LD = L.LD;
  if (e >= m) return;
  double* cov = work + (size_t)e * D * LD;
  const bool g = regularize(L, cov, D);
  __syncthreads();
  for (int q = t; q < D * D; q += kThreads) {
    const int i = q / D, j = q % D;
    covs[(size_t)e * D * D + q] = cov[i * LD + j];
    ams[(size_t)e * D * D + q] = L.AM[i * LD + j];
    axes[(size_t)e * D * D + q] = L.AX[i * LD + j];
  }
  if (t == 0) good[e] = g ? 1 : 0;
}

// ---------------------------------------------------------------------------
// Ellipsoid.__init__ from (ctr, cov) (bounding.py:201-240): eigen-decomposition
// -> axlens, axes, am, logvol.  One wavefront per ellipsoid.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
    ell_from_cov_kernel(int m, int D, const double* __restrict__ covs, double prefactor, double* axes,
                        double* axlens, double* ams, double* logvols, int* status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LD = D | 1;
  double* A = (double*)smem;
  double* V = A + D * LD;
  double* S = V + D * LD;
  double* lam = S + D * LD;
  double* rc = lam + D;
  double* rs = rc + 64;
  int* ri = (int*)(rs + 64);
  int* order = ri + 128;
  const int e = blockIdx.x, lane = threadIdx.x;
  if (e >= m) return;
  const double* C = covs + (size_t)e * D * D;
  for (int t = lane; t < D * D; t += 64) A[(t / D) * LD + t % D] = C[t];
  wave_sync();
  const bool fin = jacobi_wave(A, V, D, LD, rc, rs, ri);
  int st = DH_OK;
  if (fin) {
    sort_eigs_wave(A, V, lam, order, S, D, LD);
    double slog = 0.0;
    bool ok = true;
    for (int k = 0; k < D; ++k) {
      const double l = lam[k];
      if (!(l > 0.0) || !isfinite(l)) ok = false;
      slog += log(l);
    }
    if (!ok)
      st = DH_ERR_VALUE;
    else {
      for (int t = lane; t < D * D; t += 64) {
        const int i = t / D, j = t % D;
        axes[(size_t)e * D * D + t] = V[i * LD + j] * sqrt(lam[j]);
        double s = 0.0;
        for (int k = 0; k < D; ++k) s = fma(V[i * LD + k] * (1.0 / lam[k]), V[j * LD + k], s);
        ams[(size_t)e * D * D + t] = s;
      }
      for (int k = lane; k < D; k += 64) axlens[(size_t)e * D + k] = sqrt(lam[k]);
      if (lane == 0) logvols[e] = prefactor + 0.5 * slog;
    }
  } else {
    st = DH_ERR_VALUE;
  }
  if (lane == 0) status[e] = st;
}

// ---------------------------------------------------------------------------
// Ellipsoid.scale_to_logvol (bounding.py:242-276), one wavefront per ellipsoid.
// The eigen-system of cov is taken from the stored principal axes
// (v = axes / axlens, l = axlens^2) instead of a fresh eigh(cov).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void scale_logvol_one(int e, int m, int D, double* covs, double* ams, double* axes, double* axlens,
                                                 double* logvols, const double* __restrict__ targets, double shift,
                                                 const int* __restrict__ nells, int stride, const int* __restrict__ active,
                                                 const double* __restrict__ run_shift, unsigned char* smem) {
  double* fax = (double*)smem;  // D
  double* lax = fax + D;        // D  log axlens
  int* iso = (int*)(lax + D);
  const int lane = threadIdx.x, nt = blockDim.x;  // nt = 64 (one wavefront) or more at wide D
  auto sync = [&]() {
    if (nt > 64)
      __syncthreads();
    else
      wave_sync();
  };
  if (e >= m) return;
  // batched form: slot e belongs to run e / stride and is live iff its index
  // within the run is below that run's ellipsoid count
  if (nells && (e % stride) >= nells[e / stride]) return;
  if (active && !active[e / stride]) return;
  double* C = covs + (size_t)e * D * D;
  double* P = ams + (size_t)e * D * D;
  double* X = axes + (size_t)e * D * D;
  double* al = axlens + (size_t)e * D;
  // run_shift: a per-run addition to ln V (the bootstrap expansion); a run whose factor is 1 is left alone
  if (run_shift && run_shift[e / stride] == 0.0 && shift == 0.0) return;
  const double target = targets ? targets[e] : logvols[e] + (run_shift ? run_shift[e / stride] : 0.0) + shift;
  const double logf = target - logvols[e];
  const double max_log_axlen = log(sqrt((double)D) / 2.0);
  for (int k = lane; k < D; k += nt) lax[k] = log(al[k]);
  sync();
  if (lane == 0) {
    double mx = -INFINITY;
    for (int k = 0; k < D; ++k) mx = fmax(mx, lax[k]);
    iso[0] = (mx < max_log_axlen - logf / D) ? 1 : 0;
    if (!iso[0]) {
      // greedy per-axis inflation, largest eigenvalue first (bounding.py:262-268)
      double left = logf;
      int nleft = D;
      for (int k = 0; k < D; ++k) fax[k] = -1.0;  // marks "not done"
      for (int step = 0; step < D; ++step) {
        int best = -1;
        double bl = -INFINITY;
        // np.argsort(l)[::-1]: descending eigenvalue; ties resolve to the later index
        for (int k = 0; k < D; ++k)
          if (fax[k] < 0.0 && lax[k] >= bl) {
            bl = lax[k];
            best = k;
          }
        const double delta = fmax(fmin(max_log_axlen - lax[best], left / nleft), 0.0);
        fax[best] = exp(delta);
        left -= delta;
        --nleft;
      }
    }
  }
  sync();
  if (iso[0]) {
    const double f = exp(logf / D);
    const double f2 = f * f, inv = 1.0 / f2;
    // (round 6: every load of a thread first, then its stores -- written as `C[t] *= f2` in a loop each
    // read-modify-write waited for the one before, ten cold round trips of a wavefront: 12 us for 64 matrices)
    for (int t0 = lane; t0 < D * D; t0 += 4 * nt) {
      double c[4], p[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * nt, tc = t < D * D ? t : 0;
        c[u] = C[tc];
        p[u] = P[tc];
        x[u] = X[tc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * nt;
        if (t < D * D) {
          C[t] = c[u] * f2;
          P[t] = p[u] * inv;
          X[t] = x[u] * f;
        }
      }
    }
    for (int k = lane; k < D; k += nt) al[k] *= f;
  } else {
    // cov = (v * l1) v^T, am = (v / l1) v^T with l1 = l * fax^2, v = axes / axlens
    for (int t = lane; t < D * D; t += nt) {
      const int i = t / D, j = t % D;
      double sc = 0.0, sp = 0.0;
      for (int k = 0; k < D; ++k) {
        const double a = al[k];
        const double vik = X[i * D + k] / a, vjk = X[j * D + k] / a;
        const double l1 = a * a * fax[k] * fax[k];
        sc = fma(vik * l1, vjk, sc);
        sp = fma(vik * (1.0 / l1), vjk, sp);
      }
      C[t] = sc;
      P[t] = sp;
    }
    sync();
    for (int t = lane; t < D * D; t += nt) X[t] *= fax[t % D];
    sync();
    for (int k = lane; k < D; k += nt) al[k] *= fax[k];
  }
  if (lane == 0) logvols[e] = target;
}
// G = 0: workgroup e takes ellipsoid e.  G > 0 (the batched forms, round 6): workgroup (run, g) takes slots g, g + G, ...
// of its run -- the resident loop's enlarge launched runs x max_ells workgroups (64 x 40) for the one to three
// ellipsoids a run holds, 38 us of dispatch per rebuild.
__global__ void __launch_bounds__(1024)
    scale_logvol_kernel(int m, int D, double* covs, double* ams, double* axes, double* axlens,
                        double* logvols, const double* __restrict__ targets, double shift,
                        const int* __restrict__ nells, int stride, const int* __restrict__ active,
                        const double* __restrict__ run_shift, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (G <= 0) {
    scale_logvol_one((int)blockIdx.x, m, D, covs, ams, axes, axlens, logvols, targets, shift, nells, stride, active, run_shift, smem);
    return;
  }
  const int run = (int)blockIdx.x / G;
  const int cnt = nells ? min(nells[run], stride) : stride;
  for (int slot = (int)blockIdx.x % G; slot < cnt; slot += G) {
    scale_logvol_one(run * stride + slot, m, D, covs, ams, axes, axlens, logvols, targets, shift, nells, stride, active, run_shift, smem);
    __syncthreads();
  }
}

size_t rebuild_lds_bytes(int D, int TP = kThreads) {
  const size_t base = rebuild_lds_base_bytes(D, TP);
  const int P = (D + 1) & ~1;
  const size_t jb = 4 * (size_t)P * (P | 1) * 8;
  return base + jb > kLdsSeparate ? base : base + jb;  // else the Jacobi buffers overlay the tile
}

}  // namespace

extern "C" {

#ifdef DH_REBUILD_TIMING
void dh_rebuild_timing(long long* out16, int reset) {
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(long long));
  if (reset) {
    long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z);
  }
}
void dh_rebuild_timing_ell(long long* out128, int reset) {  // g_ell_cycles: 8 levels x 16 phases of k_ell's workgroup 0
  (void)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_ell_cycles), 128 * sizeof(long long));
  if (reset) {
    long long z[128] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ell_cycles), z, sizeof z);
  }
}
void dh_rebuild_wg_clock(long long* out, int kern, int level) {  // kWgMax x 2 wall-clock stamps (100 MHz)
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_clock), (size_t)kWgMax * 2 * sizeof(long long),
                            (((size_t)kern * 8 + level) * kWgMax * 2) * sizeof(long long));
}
void dh_rebuild_timing_levels(long long* out128, int reset) {
  (void)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_lvl_cycles), 256 * sizeof(long long));
  if (reset) {
    long long z[256] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lvl_cycles), z, sizeof z);
  }
}
#endif

// see include/dynhip.h
int dh_rebuild_batch_dev(dh_ctx* ctx, int runs, const double* pts, int n, int d, int mode, int max_ells,
                         int32_t* nells, int32_t* status, double* ctrs, double* covs, double* ams,
                         double* axes, double* axlens, double* logvols, int32_t* leaf_of_point,
                         int32_t* nnodes) {
  return dh::rebuild_launch_full(ctx, runs, pts, n, d, mode, max_ells, nells, status, ctrs, covs, ams, axes,
                                 axlens, logvols, leaf_of_point, nnodes, nullptr, nullptr);
}

int dh_rebuild_ragged_dev(dh_ctx* ctx, int runs, const double* pts, int n_max, const int32_t* n_arr, int d,
                          int mode, int max_ells, int32_t* nells, int32_t* status, double* ctrs,
                          double* covs, double* ams, double* axes, double* axlens, double* logvols) {
  return dh::rebuild_launch_full(ctx, runs, pts, n_max, d, mode, max_ells, nells, status, ctrs, covs, ams,
                                 axes, axlens, logvols, nullptr, nullptr, nullptr, n_arr);
}

}  // extern "C"

int dh::rebuild_launch_masked(dh_ctx* ctx, int runs, const double* pts, int n, int d, int mode, int max_ells,
                              int32_t* nells, int32_t* status, double* ctrs, double* covs, double* ams,
                              double* axes, double* axlens, double* logvols, const int* active) {
  return rebuild_launch_full(ctx, runs, pts, n, d, mode, max_ells, nells, status, ctrs, covs, ams, axes,
                             axlens, logvols, nullptr, nullptr, active, nullptr);
}

int dh::enlarge_launch_masked(dh_ctx* ctx, int runs, int max_ells, const int32_t* nells, int d, double* covs,
                              double* ams, double* axes, double* axlens, double* logvols,
                              double log_enlarge, const int* active, const double* run_shift) {
  const int m = runs * max_ells;
  const int G = nells ? (max_ells < 4 ? max_ells : 4) : 0;
  hipLaunchKernelGGL(scale_logvol_kernel, dim3(G > 0 ? runs * G : m), dim3(d > 64 ? 1024 : 256), (size_t)2 * d * 8 + 64, ctx->stream, m, d,
                     covs, ams, axes, axlens, logvols, (const double*)nullptr, log_enlarge, nells, max_ells,
                     active, run_shift, G);
  return hip_ok(ctx, hipGetLastError(), "enlarge launch") ? DH_OK : DH_ERR_HIP;
}

int dh::rebuild_launch_full(dh_ctx* ctx, int runs, const double* pts, int n, int d, int mode, int max_ells,
                            int32_t* nells, int32_t* status, double* ctrs, double* covs, double* ams,
                            double* axes, double* axlens, double* logvols, int32_t* leaf_of_point,
                            int32_t* nnodes, const int* active, const int* n_arr) {
  DH_CHECK_CTX(ctx);
  if (runs <= 0) return DH_OK;
  if (!pts || n < 1 || d < 1 || max_ells < 1 || (mode != 0 && mode != 1))
    return fail(ctx, DH_ERR_ARG, "rebuild: bad arguments (n=%d d=%d mode=%d)", n, d, mode);
  const size_t lds = rebuild_lds_bytes(d);
  if (lds > kLdsLimit) {
    if (mode == 1 && !active && !n_arr)
      return wide_single_launch(ctx, runs, pts, n, d, nells, status, ctrs, covs, ams, axes, axlens,
                                logvols);
    if (mode == 0 && !n_arr)  // (with a run mask: the device-resident loop's MultiEllipsoid.update, host-driven)
      return wide_multi_launch(ctx, runs, pts, n, d, max_ells, nells, status, ctrs, covs, ams, axes, axlens,
                               logvols, leaf_of_point, nnodes, active);
    if (mode == 1 && active && !n_arr)  // the device-resident loop's masked Ellipsoid.update
      return wide_single_launch_masked(ctx, runs, pts, n, d, nells, status, ctrs, covs, ams, axes, axlens, logvols,
                                       active);
    // ragged batch above d = 44 (the bootstrap replicas of a wide bound): the wide constructions take one point
    // set per call, so the sizes (and the mask) come to the host and the sets go through one after the other
    std::vector<int32_t> h_n((size_t)runs), h_act;
    if (!hip_ok(ctx, hipMemcpyAsync(h_n.data(), n_arr, (size_t)runs * 4, hipMemcpyDeviceToHost, ctx->stream), "D2H ragged sizes"))
      return DH_ERR_HIP;
    if (active) {
      h_act.resize((size_t)runs);
      if (!hip_ok(ctx, hipMemcpyAsync(h_act.data(), active, (size_t)runs * 4, hipMemcpyDeviceToHost, ctx->stream), "D2H run mask"))
        return DH_ERR_HIP;
    }
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    const size_t dd = (size_t)d * d;
    for (int s = 0; s < runs; ++s) {
      if (active && !h_act[(size_t)s]) continue;
      const int cnt = h_n[(size_t)s];
      if (cnt < 0 || cnt > n) return fail(ctx, DH_ERR_ARG, "rebuild: ragged size %d of set %d outside [0, %d]", cnt, s, n);
      const size_t o = (size_t)s * max_ells;
      int rc;
      if (cnt < 2) {  // bounding_ellipsoid of a single point raises (bounding.py:1383-1385)
        const int32_t st = DH_ERR_VALUE, one = 1;
        rc = hip_ok(ctx, hipMemcpyAsync(status + s, &st, 4, hipMemcpyHostToDevice, ctx->stream), "H2D") &&
                     hip_ok(ctx, hipMemcpyAsync(nells + s, &one, 4, hipMemcpyHostToDevice, ctx->stream), "H2D") &&
                     hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")
                 ? DH_OK
                 : DH_ERR_HIP;
      } else if (mode == 1) {
        rc = wide_single_launch(ctx, 1, pts + (size_t)s * n * d, cnt, d, nells + s, status + s, ctrs + o * d, covs + o * dd,
                                ams + o * dd, axes + o * dd, axlens + o * d, logvols + o);
      } else {
        rc = wide_multi_launch(ctx, 1, pts + (size_t)s * n * d, cnt, d, max_ells, nells + s, status + s, ctrs + o * d,
                               covs + o * dd, ams + o * dd, axes + o * dd, axlens + o * d, logvols + o,
                               leaf_of_point ? leaf_of_point + (size_t)s * n : nullptr, nnodes ? nnodes + s : nullptr);
      }
      if (rc) return rc;
    }
    return DH_OK;
  }
  RebuildArgs a;
  a.root_run0 = 0;
  a.pts = pts;
  a.n = n;
  a.d = d;
  a.runs = runs;
  a.mode = mode;
  // every split creates two children of >= 2d points each: <= n/d nodes + root
  a.max_nodes = mode == 1 ? 1 : (n / d + 3);
  a.max_ells = max_ells;
  a.prefactor = d * log(2.0) + d * lgamma(1.5) - lgamma(d / 2.0 + 1.0);
  const int LD = d | 1;
  const size_t NS = (size_t)d + 3 * (size_t)d * d + d + (size_t)d * LD;
  a.reslist_cap = a.max_nodes * 24 + 64;
  // scratch: reuse a context-owned buffer (grown on demand)
  const size_t b_perm = (size_t)runs * n * 4, b_lab = (size_t)runs * n;
  const size_t b_nodes = (size_t)runs * a.max_nodes * sizeof(Node);
  const size_t b_es = (size_t)runs * a.max_nodes * NS * 8;
  const size_t b_res = (size_t)runs * a.reslist_cap * 4;
  a.maxw = n / (4 * d) + 1;
  // depth: a balanced tree needs log2(n / 2d) levels; unbalanced splits need more.  The level kernels are launched
  // for lv levels, the work-queue form (k_tree) takes whatever is deeper.
  int lv = 4;
  while ((1 << lv) < n / (2 * d) + 1) ++lv;
  a.levels = mode == 1 ? 0 : (2 * lv + 8);
  // 128 points per k-means part: five k_split workgroups per CU (see carve_split) -- or 256 where five 256-point
  // tiles fit a CU's LDS as well (D <= 13): half the parts to meet at the device-scope barrier.  Measured (round 5, 64
  // sets, tools/r5_tps.sh): eggbox 2-D 4.48 -> 4.10 ms, two blobs 5-D 0.720 -> 0.704; at D = 25 256-point parts lose
  // (1.26 -> 1.40 ms: two workgroups per CU).  DH_SPLIT_TP overrides (64 .. 256)
  a.tps = split_lds_bytes(d, 256) * 5 <= kLdsLimit ? 256 : 128;
  if (const char* e = getenv("DH_SPLIT_TP")) {
    const int v = atoi(e);
    if (v == 64 || v == 128 || v == 192 || v == 256) a.tps = v;
  }
  a.maxp = n / a.tps + a.maxw + 1;
  // eigen-free tree nodes (MultiEllipsoid.update only: Ellipsoid.update's single node IS the output)
  a.fast = mode == 0 ? 1 : 0;
  if (const char* e = getenv("DH_REBUILD_FAST")) a.fast = a.fast && atoi(e) != 0;  // diagnostic: 0 = eigh on every node
  // The tree is built by the level pipeline (k_split / k_ell per level) for a balanced tree's depth (lv levels: an
  // idle level pair costs 10 us, and the bench trees use lv = 6 exactly);
  // whatever is deeper -- unbalanced splits -- is handed to persistent workers on a work queue (k_tree: the same
  // node routines, any depth, any node size; in the common case it finds its queue empty and leaves).
  // DH_TREE=1: the WHOLE tree by the work-queue form.  Bit-identical results (tests/test_gpu_edges.py), but
  // measured SLOWER (round 3, 20 launches each): 64 C2 runs 1.44 vs 1.29 ms, 16 runs 1.06 vs 0.95, 16 eggbox
  // runs 2.37 vs 2.15, one eggbox run 1.20 vs 0.99.  With 64 runs the chip is saturated either way (the time
  // scales with the number of workers: 256 -> 2.18 ms, 384 -> 1.68, 512 -> 1.52), and the queue form loses the
  // level pipeline's five k_split workgroups per CU (its workers carry the 77 KB layout of the ellipsoid routine:
  // two per CU), pays agent-scope (cache-bypassing) accesses for everything a node hands to the next, and leaves
  // the early parts of a multi-part node spinning until the late ones find a worker.
  a.tree = 0;
  if (const char* e = getenv("DH_TREE")) a.tree = a.fast && atoi(e) != 0;
  // level kernels for a balanced tree's depth, the work-queue tail for the rest (DH_DEEP=0: every level
  // by level kernels and no tail, as does the diagnostic slow mode; DH_DEEP_FROM=f: the tail takes over at level f)
  // (one level pair fewer -- a balanced tree's last split level is the one with n >> L >= 4 d: five pairs for the
  // bench's 2000 x 25 live sets instead of six -- was measured in round 5 and is SLOWER: 1.411 against 1.382 ms per
  // 64-run rebuild, eggbox 5.02 against 4.51: real trees are not balanced, and what is deeper than the level kernels
  // goes to the work-queue tail, which costs more than an almost idle level pair)
  int nlev = a.levels;
  if (a.fast && !(getenv("DH_DEEP") && atoi(getenv("DH_DEEP")) == 0)) nlev = a.levels < lv ? a.levels : lv;
  if (a.fast && getenv("DH_DEEP_FROM")) {
    const int f = atoi(getenv("DH_DEEP_FROM"));
    if (f >= 1 && f < a.levels) nlev = f;
  }
  if (a.tree) nlev = 0;
  const bool tail = a.fast && (a.tree || nlev < a.levels);
  a.tree_from = tail ? nlev : a.levels + 1;
  a.tq_cap = 0;
  a.kp_cap = 0;
  if (tail) {
    // partial-sum slots of the multi-part nodes the queue form may meet: a depth has at most n / tps + (nodes) parts
    a.kp_cap = a.levels * (n / a.tps + 1) + 8;
    // items: one ellipsoid per node, and per split node ceil(count / tps) parts
    a.tq_cap = runs * (2 * a.max_nodes + a.levels * (n / a.tps + 1) + 8);
  }
  const size_t lds_split = split_lds_bytes(d, a.tps);
  // the parts of one node meet at a device-scope barrier, so they must all be resident at the
  // same time: 256 parts (65 536 points per run) fit the 256 CUs with room to spare
  if ((long long)n * d >= (1ll << 31))  // (rows are addressed by 32-bit element offsets: stage_tile)
    return fail(ctx, DH_ERR_ARG, "rebuild: n x d = %lld elements per run exceeds 2^31", (long long)n * d);
  if (mode == 0 && n > 256 * kThreads)
    return fail(ctx, DH_ERR_ARG, "rebuild: MultiEllipsoid.update supports at most %d points per run (n = %d)",
                256 * kThreads, n);
  // k_finish: tree (and result list) in LDS when they fit behind the standard layout
  size_t lds_fin = lds;
  a.fin_extra_off = 0;
  a.fin_res_lds = 0;
  {
    const size_t off = (lds + 15) & ~(size_t)15;
    const size_t nb_nodes = (size_t)a.max_nodes * sizeof(Node), nb_res = (size_t)a.reslist_cap * 4;
    if (off + nb_nodes <= kLdsLimit) {
      a.fin_extra_off = (int)off;
      lds_fin = off + nb_nodes;
      if (lds_fin + nb_res <= kLdsLimit) {
        a.fin_res_lds = 1;
        lds_fin += nb_res;
      }
    }
  }
  // parts of the root: cooperative only while all parts of all runs are resident with room to spare
  // (a part idles at a barrier while part 0 runs the eigensolver, so on a full chip it only costs slots)
  // Co-residency.  Workgroups that meet at a spin barrier -- the parts of the root, the parts of one
  // k-means node -- must be on the chip together.  How many workgroups of a kernel fit is asked of the
  // runtime (occupancy API x CU count), not assumed.  The whole grid of k_root_parts must fit (every part
  // waits for part 0's solve); for k_split a CHUNK of runs must (round 5: cr runs x the level's parts per run, sized
  // to this capacity; inside a chunk the workgroups are ordered part-major so that every run starts at once): chunks
  // have consecutive workgroup ids and the dispatcher hands out workgroups in id order, so the lowest unfinished
  // chunk is always dispatched in full as the workgroups in front of it finish, and those never wait for it.
  int cap_root = 0, cap_split = 0, cap_split_level = 0, cap_tree = 0;
  {
    const int ncu = ctx->num_cu;
    int occ_root = 0, occ_split = 0;
    // (the LDS attribute must be in place for the query to count the dynamic allocation)
    (void)hipFuncSetAttribute((const void*)k_root_parts, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_root, (const void*)k_root_parts, kThreads, lds) != hipSuccess)
      occ_root = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_split, (const void*)k_split, kThreads, lds_split) != hipSuccess)
      occ_split = 1;
    cap_root = ncu * (occ_root > 0 ? occ_root : 1);
    cap_split = ncu * (occ_split > 0 ? occ_split : 1);
    cap_split_level = cap_split;  // (k_split's own: what its chunks are sized to)
    if (tail) {
      int occ_tree = 0;
      (void)hipFuncSetAttribute((const void*)k_tree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_tree, (const void*)k_tree, kThreads, lds) != hipSuccess)
        occ_tree = 1;
      cap_tree = ncu * (occ_tree > 0 ? occ_tree : 1);
      if (cap_tree < cap_split) cap_split = cap_tree;  // the parts of a node may be k_tree workgroups
    }
  }
  // The parts of a run meet at spin waits, so what is launched together must be resident together: when all runs x
  // parts do not fit, the root goes in chunks of runs that do (round 6; before, 128 runs x 8 parts fell back to the
  // single-workgroup root, 447 us against 81 us per 64 runs).
  int rp = n > 1 ? (n + kThreads - 1) / kThreads : 1;
  if (rp > cap_root || (ctx->coop_launch && (long long)runs * rp > cap_root)) rp = 1;
  int root_chunk = runs;
  if (rp > 1 && (long long)runs * rp > cap_root) root_chunk = cap_root / rp;
  if (getenv("DH_ROOT_ONE_LAUNCH") && atoi(getenv("DH_ROOT_ONE_LAUNCH")) == 1 && !ctx->coop_launch)
    root_chunk = runs;  // experiment: one launch, run-major ids, relying on in-order dispatch as k_split's chunks do
  if (getenv("DH_ROOT_CHUNK") && atoi(getenv("DH_ROOT_CHUNK")) == 0) {  // diagnostic: the form before
    root_chunk = runs;
    if ((long long)runs * rp > cap_root) rp = 1;
  }
  if (mode == 0 && (n + a.tps - 1) / a.tps > cap_split)
    return fail(ctx, DH_ERR_ARG, "rebuild: the %d parts of a %d-point node exceed the %d co-resident workgroups of k_split",
                (n + a.tps - 1) / a.tps, n, cap_split);
  if (getenv("DH_ROOT_PARTS") && atoi(getenv("DH_ROOT_PARTS")) == 0) rp = 1;  // diagnostic
  // zeroed counters: nnodes | nsplit (levels+1) | nell (levels) | nparts (levels+1) | kerr | rbar | kbar (levels x maxw)
  // ... | kp_top (runs) | tq_ctl (64) | nbar (runs x max_nodes x kBarStride) | tq_items (tq_cap x 2 ints)   [k_tree]
  const size_t n_cnt_old = (size_t)runs * ((size_t)3 * a.levels + 5 + kBarStride + (size_t)a.levels * a.maxw * kBarStride);
  const size_t n_cnt_tree = tail ? (size_t)runs + 64 + (size_t)runs * a.max_nodes * kBarStride + 2 * (size_t)a.tq_cap + 2 : 0;
  const size_t b_cnt = (n_cnt_old + n_cnt_tree) * 4;
  a.rootbuf_stride = 2 * ((size_t)rp * (2 * (size_t)d + (size_t)d * d + 1) + (size_t)d * d + 8);  // (value, tag) pairs
  const size_t b_rb = (size_t)runs * a.rootbuf_stride * 8;
  const size_t b_fl = (size_t)runs * a.max_nodes * 8, b_fi = (size_t)runs * a.max_nodes * 2 * 4;
  const size_t b_pl = (size_t)2 * runs * a.maxp * 2 * 4, b_pb = (size_t)2 * runs * a.maxw * 4;
  const size_t b_kp = mode == 1 ? 0 : (size_t)2 * runs * a.maxp * (2 * (size_t)d + 2) * 16;  // (value, tag) pairs
  const size_t b_kpt = tail ? (size_t)2 * runs * a.kp_cap * (2 * (size_t)d + 2) * 16 : 0;  // the queue form's own
  const size_t b_sl = (size_t)2 * runs * a.maxw * 4, b_el = (size_t)(a.levels > 0 ? a.levels : 1) * runs * 2 * a.maxw * 4;
  const size_t b_sc = (size_t)runs * d * 8;
  const size_t b_ps = mode == 1 ? 0 : (size_t)runs * n * d * 8;
  const size_t b_of = a.fast ? (size_t)runs * max_ells * 4 : 0;
  const size_t b_re = a.fast ? (size_t)runs * (2 * (size_t)d * d + d + 2) * 8 : 0;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t total = al(b_perm) * 2 + al(b_lab) + al(b_nodes) + al(b_es) + al(b_res) + al(b_cnt) +
                       al(b_sl) + al(b_el) + al(b_sc) + al(b_ps) + al(b_pl) + al(b_pb) + al(b_kp) + al(b_kpt) + al(b_rb) + al(b_fl) + al(b_fi) +
                       2 * al(b_of) + al(b_re);
  if (total > ctx->rebuild_ws_cap) {
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync")) return DH_ERR_HIP;
    if (ctx->rebuild_ws) (void)hipFree(ctx->rebuild_ws);
    ctx->rebuild_ws = nullptr;
    ctx->rebuild_ws_cap = 0;
    if (!hip_ok(ctx, hipMalloc((void**)&ctx->rebuild_ws, total), "hipMalloc(rebuild scratch)"))
      return DH_ERR_NOMEM;
    ctx->rebuild_ws_cap = total;
    // (the k-means partials are recognised by their tags: no stale word of a fresh allocation may pass for one)
    if (!hip_ok(ctx, hipMemsetAsync(ctx->rebuild_ws, 0, total, ctx->stream), "memset(rebuild scratch)")) return DH_ERR_HIP;
  }
  char* w = ctx->rebuild_ws;
  a.perm = (int*)w;
  w += al(b_perm);
  a.perm2 = (int*)w;
  w += al(b_perm);
  a.lab = (unsigned char*)w;
  w += al(b_lab);
  a.nodes = (Node*)w;
  w += al(b_nodes);
  a.estore = (double*)w;
  w += al(b_es);
  a.reslist = (int*)w;
  w += al(b_res);
  int* cnt = (int*)w;
  w += al(b_cnt);
  a.nnodes_dev = cnt;
  a.nsplit = cnt + runs;
  a.nell = a.nsplit + (size_t)(a.levels + 1) * runs;
  a.nparts = a.nell + (size_t)a.levels * runs;
  a.kerr = a.nparts + (size_t)(a.levels + 1) * runs;
  a.rbar = a.kerr + runs;
  a.kbar = a.rbar + (size_t)runs * kBarStride;
  a.kp_top = a.nbar = a.tq_ctl = nullptr;
  a.tq_items = nullptr;
  if (tail) {
    a.kp_top = cnt + n_cnt_old;
    a.tq_ctl = a.kp_top + runs;
    a.nbar = a.tq_ctl + 64;
    int* q = a.nbar + (size_t)runs * a.max_nodes * kBarStride;
    if (((uintptr_t)q) & 7) ++q;  // 8-byte items
    a.tq_items = (unsigned long long*)q;
  }
  a.split_list = (int*)w;
  w += al(b_sl);
  a.ell_list = (int*)w;
  w += al(b_el);
  a.scale_g = (double*)w;
  w += al(b_sc);
  a.pts_scaled = (double*)w;
  w += al(b_ps);
  a.part_list = (int*)w;
  w += al(b_pl);
  a.part_base = (int*)w;
  w += al(b_pb);
  a.kpart = (double*)w;
  w += al(b_kp);
  double* kpart_tail = (double*)w;
  w += al(b_kpt);
  a.rootbuf = (double*)w;
  w += al(b_rb);
  a.fin_lse = (double*)w;
  w += al(b_fl);
  a.fin_int = (int*)w;
  w += al(b_fi);
  a.out_node = a.out_fast = nullptr;
  a.root_eig = nullptr;
  if (a.fast) {
    a.out_node = (int*)w;
    w += al(b_of);
    a.out_fast = (int*)w;
    w += al(b_of);
    a.root_eig = (double*)w;
    w += al(b_re);
  }
  a.nells = nells;
  a.status = status;
  a.ctrs = ctrs;
  a.covs = covs;
  a.ams = ams;
  a.axes = axes;
  a.axlens = axlens;
  a.logvols = logvols;
  a.leaf_of_point = leaf_of_point;
  a.nnodes_out = nnodes;
  a.active = active;
  a.n_arr = n_arr;
  a.epoch = ++ctx->rebuild_epoch;
  // k_ell's top levels: a tile of 512 points, if it fits (D <= 30); DH_ELL_TOP_TILE=0: off
  size_t lds_top = rebuild_lds_bytes(d, 2 * kThreads);
  if (lds_top > kLdsLimit || mode != 0 || (getenv("DH_ELL_TOP_TILE") && atoi(getenv("DH_ELL_TOP_TILE")) == 0)) lds_top = 0;
  DH_DEV_MEMO(attr_lds);
  DH_DEV_MEMO(attr_top);
  if (lds > attr_lds) {
    const void* ks[8] = {(const void*)k_root_parts, (const void*)k_split, (const void*)k_ell<false>,
                         (const void*)k_ell<true>, (const void*)k_out_eig, (const void*)k_root_eig,
                         (const void*)k_tree, (const void*)k_ell<false, true>};
    for (const void* kf : ks)
      if (!hip_ok(ctx, hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                  "hipFuncSetAttribute(rebuild LDS)"))
        return DH_ERR_HIP;
    attr_lds = lds;
    attr_top = 0;  // (k_ell's limit was just lowered to lds)
  }
  if (lds_top > attr_top) {
    if (!hip_ok(ctx, hipFuncSetAttribute((const void*)k_ell<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_top),
                "hipFuncSetAttribute(k_ell LDS)") ||
        !hip_ok(ctx, hipFuncSetAttribute((const void*)k_ell<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_top),
                "hipFuncSetAttribute(k_ell LDS)") ||
        !hip_ok(ctx, hipFuncSetAttribute((const void*)k_ell<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_top),
                "hipFuncSetAttribute(k_ell LDS)"))
      return DH_ERR_HIP;
    attr_top = lds_top;
  }
  DH_DEV_MEMO(attr_fin);
  if (lds_fin > attr_fin) {
    if (!hip_ok(ctx, hipFuncSetAttribute((const void*)k_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fin),
                "hipFuncSetAttribute(k_finish LDS)"))
      return DH_ERR_HIP;
    attr_fin = lds_fin;
  }
  if (!hip_ok(ctx, hipMemsetAsync(cnt, 0, b_cnt, ctx->stream), "memset(rebuild counters)")) return DH_ERR_HIP;
  for (int r0 = 0; r0 < runs; r0 += root_chunk) {
    a.root_run0 = r0;
    const int cr = runs - r0 < root_chunk ? runs - r0 : root_chunk;
    if (!hip_ok(ctx, launch_all_resident(ctx, k_root_parts, dim3(cr * rp), dim3(kThreads), lds, a, rp), "k_root_parts launch"))
      return DH_ERR_HIP;
  }
  a.root_run0 = 0;
  bool forked = false;
  if (a.fast && !(getenv("DH_ROOT_EIG_SIDE") && atoi(getenv("DH_ROOT_EIG_SIDE")) == 0)) {
    if (!ctx->side_stream) {
      // lowest priority: what runs here (the root's eigen-system, the leaves) is off the critical path and must not
      // take workgroup slots / LDS from the level kernels that are ready at the same moment
      int pr_lo = 0, pr_hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
      if (getenv("DH_SIDE_PRIO") && atoi(getenv("DH_SIDE_PRIO")) == 0) pr_lo = 0;
      if (!hip_ok(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, pr_lo), "hipStreamCreate(side)") ||
          !hip_ok(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming), "hipEventCreate") ||
          !hip_ok(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming), "hipEventCreate"))
        return DH_ERR_HIP;
    }
    if (!hip_ok(ctx, hipEventRecord(ctx->ev_fork, ctx->stream), "hipEventRecord(fork)") ||
        !hip_ok(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)"))
      return DH_ERR_HIP;
    hipLaunchKernelGGL(k_root_eig, dim3(runs), dim3(kThreads), lds, ctx->side_stream, a);
    if (!hip_ok(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream), "hipEventRecord(join)")) return DH_ERR_HIP;
    forked = true;
  } else {
    a.root_eig = nullptr;
  }
  // small nodes by one wavefront each (k_ell_wave), from the level where the average child fits -- where eight such
  // nodes (128-point tile, four D x D matrices) share a CU's LDS: D <= 13.  Measured (tools/wave_ell_ab.py, 64 sets):
  // eggbox 2-D 6.54 -> 4.67 ms, two blobs 5-D 0.91 -> 0.75, 3-D blob 0.71 -> 0.62.  At D = 25 the leaves-only form
  // LOSES (1.29 -> 1.41 ms: 39 point loads and 13 matrix rows per lane and sweep outweigh the rounds saved), so
  // above D = 13 it is off.  DH_WAVE_ELL=0: off; =1: leaves only; =2: with the axis, at any D that fits 64 KB.
  int wave_from = nlev, wave_cap = 0, wave_axis = 0;
  size_t lds_wave = 0;
  if (a.fast) {
    const char* e = getenv("DH_WAVE_ELL");
    const int mode_w = e ? atoi(e) : -1;
    if (mode_w != 0) {
      wave_cap = 128;
      const bool fits8 = wave_lds_bytes(d, wave_cap, true) * 8 <= kLdsLimit;
      wave_axis = mode_w == 2 || (mode_w < 0 && fits8) ? 1 : 0;
      if (!wave_axis) wave_cap = 4 * d - 1 < 128 ? 4 * d - 1 : 128;
      lds_wave = wave_lds_bytes(d, wave_cap, wave_axis != 0);
      if ((mode_w > 0 || fits8) && wave_cap >= 2 && lds_wave <= 64 * 1024) {
        wave_from = 0;
        while (wave_from < nlev && (n >> (wave_from + 1)) > 2 * wave_cap) ++wave_from;
      }
    }
  }
  // Leaves beside the tree (round 5).  A child too small to be split again (count < 4 d) is read by nothing but
  // k_finish, yet the level kernels built it on the critical path: the 2 048 leaves of the 64-run bench tree's last busy
  // level are four rounds of k_ell's 512 workgroup slots (150 us).  Above D = 13 (below, k_ell_wave's one-wavefront form
  // with the axis serves on the main stream) the leaves of the levels whose average child is leaf-sized go to
  // k_ell_wave<128> on the SIDE stream -- two wavefronts and 31 KB of LDS a node: five per CU -- while the main stream
  // goes on with the level's few splittable children and the next level pair; k_ell skips exactly the nodes that kernel
  // takes.  The side stream joins before the work-queue tail, to which a declined leaf (eigen-free path not applicable)
  // is queued.  Same routines, same bits (tests/test_gpu_edges.py).  Round 6: with the level kernels a third shorter the
  // side stream no longer pays -- measured on the bench shard (tools/r6_env.sh): 64 runs 1.065 ms either way, one run
  // 0.70 -> 0.66 and 128 runs 2.05 -> 2.01 WITHOUT it (two more event waits per level, five 31 KB leaf workgroups per
  // CU beside the level's own) -- so it is off by default; DH_LEAF_SIDE=1 switches it on.
  // Round 6, second half: the leaves by a light kernel of their own on the MAIN stream (DH_LEAF_MAIN=1; measured and
  // left off).  A leaf needs neither the axis nor a second tile: k_ell_wave<256> -- the same routines with four
  // wavefronts, a carve of the tile (4 d - 1 points) and two matrices, 31 KB instead of k_ell's 77 -- in front of the
  // level's k_ell, which skips what it takes.  The 2 048 leaves of the 64-run bench tree's last busy level are four
  // rounds of k_ell's two workgroups per CU (112 us); the light kernel took 90 us for them (its wave-by-wave fold, 20
  // spilled registers at the 168 of three workgroups per CU) and k_ell another 56 for the level's splittable children:
  // 1.04 -> 1.09 ms per 64 runs.
  int leaf_from = nlev, leaf_cap = 0;
  size_t lds_leaf = 0;
  const bool leaf_side = getenv("DH_LEAF_SIDE") && atoi(getenv("DH_LEAF_SIDE")) == 1;
  const bool leaf_main = !leaf_side && getenv("DH_LEAF_MAIN") && atoi(getenv("DH_LEAF_MAIN")) == 1;
  if (a.fast && mode == 0 && tail && (forked || leaf_main) && wave_from >= nlev && d >= 14 && (leaf_side || leaf_main)) {
    leaf_cap = 4 * d - 1 < 128 ? 4 * d - 1 : 128;
    lds_leaf = wave_lds_bytes(d, leaf_cap, false);
    if (lds_leaf <= 64 * 1024) {
      // from the level whose average child is below 3 d points ... (n >> (L + 1)) < 4 d would start a level earlier,
      // where most children are still splittable
      leaf_from = 0;
      int lf = 3 * d;
      if (const char* e = getenv("DH_LEAF_FROM_PTS")) lf = atoi(e) > 0 ? atoi(e) : lf;
      while (leaf_from < nlev && (n >> (leaf_from + 1)) >= lf) ++leaf_from;
      if (!ctx->ev_leaf && !hip_ok(ctx, hipEventCreateWithFlags(&ctx->ev_leaf, hipEventDisableTiming), "hipEventCreate"))
        return DH_ERR_HIP;
    } else {
      leaf_cap = 0;
    }
  }
  bool side_leaves = false;
  for (int L = 0; L < nlev; ++L) {
    // Grids no larger than the level can need (round 5): level L splits at most 2^L nodes of a run -- at most
    // n / tps + 2^L parts -- and creates at most 2^(L + 1) children.  Workgroups are dispatched at a finite rate: the
    // 2 368 / 2 688-workgroup grids of the worst case cost the first levels 60 us each at 64 runs, most of them for
    // workgroups that found nothing to do.
    const long long nodes_L = L < 20 ? (1ll << L) : (1ll << 20);
    const int gp = (int)(a.maxp < (long long)n / a.tps + nodes_L + 1 ? a.maxp : (long long)n / a.tps + nodes_L + 1);
    const int ge = (int)(2ll * a.maxw < 2 * nodes_L ? 2ll * a.maxw : 2 * nodes_L);
    // runs per chunk: cr * gp workgroups resident together, with an eighth of the chip to spare (the side stream's
    // kernels hold slots too; a chunk that does not fit would still finish -- they do not wait for it -- only later)
    // (ONE context per GPU is assumed, as for k_root_parts: a second process -- or a long-lived foreign kernel -- can
    // hold slots this sizing counts on; DH_SPLIT_RESIDENT_PCT lowers the share of the chip a chunk may claim (87 by
    // default, e.g. 40 on a GPU shared by two processes), and the spin limit fails a starved run instead of hanging)
    static const int resident_pct = [] {
      const char* e = getenv("DH_SPLIT_RESIDENT_PCT");
      const int v = e ? atoi(e) : 0;
      return v >= 1 && v <= 100 ? v : 87;
    }();
    // (round 6) no more k_ell / k_ell_wave workgroups than a few rounds of the chip: a workgroup takes every ge-th child
    // of its run (the kernels' own loops).  The bound above is the worst case; a many-mode tree's deep level (eggbox 2-D,
    // nlive 5 000, 16 runs: 1 065 parts and 1 252 children possible per run, some 200 there) was 20 000 workgroups of
    // which a fifth found work, and the dispatch of the rest half the level's time.
    const int grid_cap = getenv("DH_LEVEL_GRID_CAP") ? atoi(getenv("DH_LEVEL_GRID_CAP")) : 1;
    const int split_room = cap_split_level > 0 ? (int)((long long)cap_split_level * resident_pct / 100) : 1;
    const int gp_l = gp;  // (k_split keeps the worst case: a loop over parts in it costs registers it does not have --
                          // 5 spilled VGPRs -- and, where the parts are real, serialises two k-means chains)
    int ge_l = ge, gw_l = ge;
    if (grid_cap > 0) {
      const int cap_ell = 2 * ctx->num_cu;  // (k_ell: two workgroups per CU)
      const int want_e = 8 * cap_ell / runs > 1 ? 8 * cap_ell / runs : 1;
      if (want_e < ge_l) ge_l = want_e;
      // (k_ell_wave: 16 384 one-wavefront workgroups; 8 192 / 4 096 / 2 048 measured on the C3 loop: 0.088 / 0.089 / 0.089 s
      // against 0.087 -- its time is its nodes, not its dispatch)
      const int want_w = 16384 / runs > 1 ? 16384 / runs : 1;
      if (want_w < gw_l) gw_l = want_w;
    }
    int cr = cap_split_level > 0 ? split_room / gp_l : 1;
    cr = cr < 1 ? 1 : (cr > runs ? runs : cr);
    const int nchunk = (runs + cr - 1) / cr;
    cr = (runs + nchunk - 1) / nchunk;  // (chunks of equal size)
    hipLaunchKernelGGL(k_split, dim3(nchunk * cr * gp_l), dim3(kThreads), lds_split, ctx->stream, a, L, gp_l, cr);
    const int wave = L >= wave_from ? 1 : 0;
    if (L >= wave_from)
      hipLaunchKernelGGL(k_ell_wave<64>, dim3(runs * gw_l), dim3(64), lds_wave, ctx->stream, a, L, gw_l,
                         wave_cap, wave_axis, 0);
    const int lc = (leaf_cap > 0 && L >= leaf_from) ? leaf_cap : 0;
    if (lc && leaf_main)
      hipLaunchKernelGGL(k_ell_wave<256>, dim3(runs * ge), dim3(256), lds_leaf, ctx->stream, a, L, ge, leaf_cap, 0, 1);
    if (lc && leaf_side && !hip_ok(ctx, hipEventRecord(ctx->ev_leaf, ctx->stream), "hipEventRecord(leaf fork)")) return DH_ERR_HIP;
    // The top levels' children are several 256-point tiles each and few (one workgroup per CU or less): their
    // workgroups stage 512 points at once -- a 1 000-point child is gathered three times instead of seven (covariance
    // pass 2 + Mahalanobis pass 1, the last tile still staged), a 500-point child once instead of three times.  Only
    // while the level's workgroups all fit the chip at one per CU (the LDS of such a tile allows no second one).
    // (round 6: only while the level's workgroups fill at most HALF the CUs -- at 128 runs level 0's 256 one-per-CU
    // workgroups with the big tile lost to two-per-CU with the small one: 2.09 -> 2.05 ms; 64 runs unchanged)
    static const int top_div = getenv("DH_ELL_TOP_DIV") ? atoi(getenv("DH_ELL_TOP_DIV")) : 2;
    const bool top = lds_top > 0 && (n >> (L + 1)) > kThreads && (long long)runs * ge * (top_div > 0 ? top_div : 1) <= ctx->num_cu;
    if (a.fast && tail && !(getenv("DH_ELL_DEFER") && atoi(getenv("DH_ELL_DEFER")) == 0))
      hipLaunchKernelGGL((k_ell<false, true>), dim3(runs * ge_l), dim3(kThreads), top ? lds_top : lds, ctx->stream, a, L, ge_l, wave, lc,
                         top ? 2 * kThreads : kThreads);
    else if (a.fast)
      hipLaunchKernelGGL(k_ell<false>, dim3(runs * ge_l), dim3(kThreads), top ? lds_top : lds, ctx->stream, a, L, ge_l, wave, lc,
                         top ? 2 * kThreads : kThreads);
    else
      hipLaunchKernelGGL(k_ell<true>, dim3(runs * ge), dim3(kThreads), top ? lds_top : lds, ctx->stream, a, L, ge, 0, 0,
                         top ? 2 * kThreads : kThreads);
    if (lc && leaf_side) {  // (submitted after the level's k_ell: its few splittable children should get their slots first)
      if (!hip_ok(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_leaf, 0), "hipStreamWaitEvent(leaf fork)"))
        return DH_ERR_HIP;
      hipLaunchKernelGGL(k_ell_wave<128>, dim3(runs * ge), dim3(128), lds_leaf, ctx->side_stream, a, L, ge,
                         leaf_cap, 0, 1);
      side_leaves = true;
    }
  }
  if (side_leaves) {  // the join moves behind the leaves (k_root_eig is long done) and in front of the tail
    if (!hip_ok(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream), "hipEventRecord(join)") ||
        !hip_ok(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "hipStreamWaitEvent(join)"))
      return DH_ERR_HIP;
    forked = false;
  }
  if (tail) {
    // persistent workers: as many as can be resident (the parts of a node meet at spin barriers), but
    // no more than the tree can ever keep busy; the queue form indexes its own partial-sum slots
    RebuildArgs at = a;
    at.tree = 1;
    at.kpart = kpart_tail;
    at.maxp = a.kp_cap;
    long long want = (long long)runs * (n / a.tps + 2 * a.maxw + 1);
    int G = (int)(want < cap_tree ? (want < 1 ? 1 : want) : cap_tree);
    if (const char* e = getenv("DH_TREE_G")) G = atoi(e) > 0 ? atoi(e) : G;
    at.tq_sleep = getenv("DH_TREE_SLEEP") ? atoi(getenv("DH_TREE_SLEEP")) : 1;
    at.tq_nocoh = getenv("DH_TREE_NOCOH") ? atoi(getenv("DH_TREE_NOCOH")) : 0;
    if (getenv("DH_TREE_STATS")) {  // diagnostic: what the level kernels left for the work-queue tail (a sync per rebuild)
      int q[64];
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipMemcpy(q, at.tq_ctl, sizeof q, hipMemcpyDeviceToHost);
      const int nq = q[16] > 0 ? (q[16] < a.tq_cap ? q[16] : a.tq_cap) : 0;
      std::vector<unsigned long long> items((size_t)nq + 1);
      if (nq > 0) (void)hipMemcpy(items.data(), at.tq_items, (size_t)nq * 8, hipMemcpyDeviceToHost);
      int nsplit = 0, nell = 0;
      for (int i = 0; i < nq; ++i) (items[i] & kItemSplit ? nsplit : nell) += 1;
      {  // sizes and depths of the declined nodes
        std::vector<Node> nd((size_t)runs * a.max_nodes);
        (void)hipMemcpy(nd.data(), a.nodes, nd.size() * sizeof(Node), hipMemcpyDeviceToHost);
        int hist_depth[16] = {0}, cmin = 1 << 30, cmax = 0;
        long long csum = 0;
        for (int i = 0; i < nq; ++i)
          if (!(items[i] & kItemSplit)) {
            const int r = (int)((items[i] >> 40) & 0x3fffff), nn = (int)((items[i] >> 16) & 0xffffff);
            const Node& x = nd[(size_t)r * a.max_nodes + nn];
            hist_depth[x.depth < 15 ? x.depth : 15] += 1;
            cmin = x.count < cmin ? x.count : cmin;
            cmax = x.count > cmax ? x.count : cmax;
            csum += x.count;
          }
        if (nell)
          fprintf(stderr, "  declined nodes: count %d .. %d (mean %.0f); by depth 1..6: %d %d %d %d %d %d\n", cmin, cmax, (double)csum / nell,
                  hist_depth[1], hist_depth[2], hist_depth[3], hist_depth[4], hist_depth[5], hist_depth[6]);
      }
      fprintf(stderr, "rebuild tail: %d items queued by the level kernels (%d runs, n %d, d %d, %d levels): %d ellipsoids (declined by the "
              "eigen-free path) + %d k-means parts (deeper than the level plan)\n", nq, runs, n, d, nlev, nell, nsplit);
    }
    hipLaunchKernelGGL(k_tree, dim3(G), dim3(kThreads), lds, ctx->stream, at);
  }
  if (forked && !hip_ok(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "hipStreamWaitEvent(join)")) return DH_ERR_HIP;
  hipLaunchKernelGGL(k_finish, dim3(runs), dim3(kThreads), lds_fin, ctx->stream, a);
  if (a.fast) {
    const int G = max_ells < 8 ? max_ells : 8;
    hipLaunchKernelGGL(k_out_eig, dim3(runs * G), dim3(kThreads), lds, ctx->stream, a, G);
  }
  return hip_ok(ctx, hipGetLastError(), "rebuild launch") ? DH_OK : DH_ERR_HIP;
}

extern "C" {

int dh_rebuild(dh_ctx* ctx, const double* pts, int n, int d, int mode, int max_ells, int32_t* nells,
               double* ctrs, double* covs, double* ams, double* axes, double* axlens, double* logvols,
               int32_t* leaf_of_point, int32_t* nnodes) {
  DH_CHECK_CTX(ctx);
  if (!pts || !nells || !ctrs || !covs || !ams || !axes || !axlens || !logvols)
    return fail(ctx, DH_ERR_ARG, "rebuild: null pointer");
  arena_reset(ctx);
  const size_t dd = (size_t)d * d;
  const size_t need = ((size_t)n * d + (size_t)max_ells * (3 * dd + 2 * d + 1)) * 8 + (size_t)n * 4 + 8192;
  int rc = arena_reserve(ctx, need);
  if (rc) return rc;
  const double* d_pts = arena_up(ctx, pts, (size_t)n * d);
  int32_t* d_nells = (int32_t*)arena_get(ctx, 4);
  int32_t* d_status = (int32_t*)arena_get(ctx, 4);
  int32_t* d_nn = (int32_t*)arena_get(ctx, 4);
  double* d_ctrs = (double*)arena_get(ctx, (size_t)max_ells * d * 8);
  double* d_covs = (double*)arena_get(ctx, (size_t)max_ells * dd * 8);
  double* d_ams = (double*)arena_get(ctx, (size_t)max_ells * dd * 8);
  double* d_axes = (double*)arena_get(ctx, (size_t)max_ells * dd * 8);
  double* d_axl = (double*)arena_get(ctx, (size_t)max_ells * d * 8);
  double* d_lv = (double*)arena_get(ctx, (size_t)max_ells * 8);
  int32_t* d_lop = leaf_of_point ? (int32_t*)arena_get(ctx, (size_t)n * 4) : nullptr;
  if (!d_pts || !d_nells || !d_status || !d_nn || !d_ctrs || !d_covs || !d_ams || !d_axes || !d_axl ||
      !d_lv || (leaf_of_point && !d_lop))
    return DH_ERR_NOMEM;
  rc = dh_rebuild_batch_dev(ctx, 1, d_pts, n, d, mode, max_ells, d_nells, d_status, d_ctrs, d_covs, d_ams,
                            d_axes, d_axl, d_lv, d_lop, d_nn);
  if (rc) return rc;
  int32_t h_status = 0, h_nells = 0;
  if (!down(ctx, &h_status, d_status, 1) || !down(ctx, &h_nells, d_nells, 1)) return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  if (h_status != DH_OK) {
    const char* msg = h_status == DH_ERR_VALUE     ? "Cannot compute a bounding ellipsoid (single point or singular covariance)"
                      : h_status == DH_ERR_CONTAIN ? "Failed to initialize the ellipsoid to contain all the points"
                      : h_status == DH_ERR_REGION  ? "Rejecting invalid MultiEllipsoid region"
                      : h_status == DH_ERR_NOMEM   ? "rebuild: more ellipsoids/nodes than the output buffers hold"
                                                   : "rebuild failed";
    return fail(ctx, h_status, "%s", msg);
  }
  *nells = h_nells;
  const size_t m = (size_t)h_nells;
  if (!down(ctx, ctrs, d_ctrs, m * d) || !down(ctx, covs, d_covs, m * dd) || !down(ctx, ams, d_ams, m * dd) ||
      !down(ctx, axes, d_axes, m * dd) || !down(ctx, axlens, d_axl, m * d) || !down(ctx, logvols, d_lv, m) ||
      !down(ctx, leaf_of_point, d_lop, (size_t)n) || !down(ctx, nnodes, d_nn, 1))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_ell_from_cov(dh_ctx* ctx, int m, int d, const double* covs, double* axes, double* axlens,
                    double* ams, double* logvols) {
  DH_CHECK_CTX(ctx);
  if (m <= 0) return DH_OK;
  if (!covs || !axes || !axlens || !ams || !logvols || d < 1)
    return fail(ctx, DH_ERR_ARG, "ell_from_cov: bad arguments");
  const int LD = d | 1;
  const size_t lds = ((size_t)3 * d * LD + d + 128) * 8 + (128 + (size_t)d + 8) * 4;
  if (lds > 160 * 1024) return fail(ctx, DH_ERR_ARG, "ell_from_cov: d=%d too large for LDS", d);
  arena_reset(ctx);
  const size_t dd = (size_t)d * d;
  int rc = arena_reserve(ctx, (size_t)m * (3 * dd + d + 2) * 8 + 8192);
  if (rc) return rc;
  const double* d_c = arena_up(ctx, covs, (size_t)m * dd);
  double* d_ax = (double*)arena_get(ctx, (size_t)m * dd * 8);
  double* d_am = (double*)arena_get(ctx, (size_t)m * dd * 8);
  double* d_al = (double*)arena_get(ctx, (size_t)m * d * 8);
  double* d_lv = (double*)arena_get(ctx, (size_t)m * 8);
  int* d_st = (int*)arena_get(ctx, (size_t)m * 4);
  if (!d_c || !d_ax || !d_am || !d_al || !d_lv || !d_st) return DH_ERR_NOMEM;
  const double pre = d * log(2.0) + d * lgamma(1.5) - lgamma(d / 2.0 + 1.0);
  DH_DEV_MEMO(attr_lds);
  if (lds > attr_lds) {
    if (!hip_ok(ctx,
                hipFuncSetAttribute((const void*)ell_from_cov_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "hipFuncSetAttribute(ell_from_cov LDS)"))
      return DH_ERR_HIP;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(ell_from_cov_kernel, dim3(m), dim3(64), lds, ctx->stream, m, d, d_c, pre, d_ax, d_al,
                     d_am, d_lv, d_st);
  if (!hip_ok(ctx, hipGetLastError(), "ell_from_cov launch")) return DH_ERR_HIP;
  std::vector<int> st((size_t)m);
  if (!down(ctx, axes, d_ax, (size_t)m * dd) || !down(ctx, ams, d_am, (size_t)m * dd) ||
      !down(ctx, axlens, d_al, (size_t)m * d) || !down(ctx, logvols, d_lv, (size_t)m) ||
      !down(ctx, st.data(), d_st, (size_t)m))
    return DH_ERR_HIP;
  if ((rc = dh_sync(ctx))) return rc;
  for (int i = 0; i < m; ++i)
    if (st[i] != DH_OK)
      return fail(ctx, DH_ERR_VALUE, "The input covariance of ellipsoid %d is singular or not finite", i);
  return DH_OK;
}

int dh_improve_covar_mat(dh_ctx* ctx, int m, int d, const double* covs_in, int32_t* good, double* covs,
                         double* ams, double* axes) {
  DH_CHECK_CTX(ctx);
  if (m <= 0) return DH_OK;
  if (!covs_in || !good || !covs || !ams || !axes || d < 1)
    return fail(ctx, DH_ERR_ARG, "improve_covar_mat: bad arguments");
  if (d > 44) return fail(ctx, DH_ERR_ARG, "improve_covar_mat: d=%d > 44 (narrow-D routine)", d);
  const int LD = d | 1;
  const size_t lds = rebuild_lds_bytes(d);
  const size_t dd = (size_t)d * d;
  arena_reset(ctx);
  int rc = arena_reserve(ctx, (size_t)m * ((size_t)d * LD + 3 * dd + 1) * 8 + 8192);
  if (rc) return rc;
  std::vector<double> padded((size_t)m * d * LD, 0.0);
  for (int e = 0; e < m; ++e)
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) padded[((size_t)e * d + i) * LD + j] = covs_in[(size_t)e * dd + i * d + j];
  double* d_w = arena_up(ctx, padded.data(), padded.size());
  double* d_c = (double*)arena_get(ctx, (size_t)m * dd * 8);
  double* d_am = (double*)arena_get(ctx, (size_t)m * dd * 8);
  double* d_ax = (double*)arena_get(ctx, (size_t)m * dd * 8);
  int* d_g = (int*)arena_get(ctx, (size_t)m * 4);
  if (!d_w || !d_c || !d_am || !d_ax || !d_g) return DH_ERR_NOMEM;
  if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync(H2D of a temporary)")) return DH_ERR_HIP;
  DH_DEV_MEMO(attr_icm);
  if (lds > attr_icm) {
    if (!hip_ok(ctx, hipFuncSetAttribute((const void*)improve_cov_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "hipFuncSetAttribute(improve_cov LDS)"))
      return DH_ERR_HIP;
    attr_icm = lds;
  }
  hipLaunchKernelGGL(improve_cov_kernel, dim3(m), dim3(kThreads), lds, ctx->stream, m, d, d_w, d_g, d_c, d_am, d_ax);
  if (!hip_ok(ctx, hipGetLastError(), "improve_covar_mat launch")) return DH_ERR_HIP;
  if (!down(ctx, covs, d_c, (size_t)m * dd) || !down(ctx, ams, d_am, (size_t)m * dd) ||
      !down(ctx, axes, d_ax, (size_t)m * dd) || !down(ctx, (int*)good, d_g, (size_t)m))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

int dh_enlarge_batch_dev(dh_ctx* ctx, int runs, int max_ells, const int32_t* nells, int d, double* covs,
                         double* ams, double* axes, double* axlens, double* logvols, double log_enlarge) {
  DH_CHECK_CTX(ctx);
  if (runs <= 0) return DH_OK;
  if (!nells || !covs || !ams || !axes || !axlens || !logvols || d < 1 || max_ells < 1)
    return fail(ctx, DH_ERR_ARG, "enlarge: bad arguments");
  const int m = runs * max_ells;
  const int G = max_ells < 4 ? max_ells : 4;
  hipLaunchKernelGGL(scale_logvol_kernel, dim3(runs * G), dim3(d > 64 ? 1024 : 256), (size_t)2 * d * 8 + 64, ctx->stream, m, d,
                     covs, ams, axes, axlens, logvols, (const double*)nullptr, log_enlarge, nells,
                     max_ells, (const int*)nullptr, (const double*)nullptr, G);
  return hip_ok(ctx, hipGetLastError(), "enlarge launch") ? DH_OK : DH_ERR_HIP;
}

int dh_scale_to_logvol(dh_ctx* ctx, int m, int d, double* covs, double* ams, double* axes, double* axlens,
                       double* logvols, const double* targets) {
  DH_CHECK_CTX(ctx);
  if (m <= 0) return DH_OK;
  if (!covs || !ams || !axes || !axlens || !logvols || !targets || d < 1)
    return fail(ctx, DH_ERR_ARG, "scale_to_logvol: bad arguments");
  arena_reset(ctx);
  const size_t dd = (size_t)d * d;
  int rc = arena_reserve(ctx, (size_t)m * (3 * dd + d + 2) * 8 + 8192);
  if (rc) return rc;
  double* d_c = arena_up(ctx, (const double*)covs, (size_t)m * dd);
  double* d_p = arena_up(ctx, (const double*)ams, (size_t)m * dd);
  double* d_x = arena_up(ctx, (const double*)axes, (size_t)m * dd);
  double* d_al = arena_up(ctx, (const double*)axlens, (size_t)m * d);
  double* d_lv = arena_up(ctx, (const double*)logvols, (size_t)m);
  const double* d_t = arena_up(ctx, targets, (size_t)m);
  if (!d_c || !d_p || !d_x || !d_al || !d_lv || !d_t) return DH_ERR_NOMEM;
  hipLaunchKernelGGL(scale_logvol_kernel, dim3(m), dim3(d > 64 ? 1024 : 256), (size_t)2 * d * 8 + 64, ctx->stream, m, d,
                     d_c, d_p, d_x, d_al, d_lv, d_t, 0.0, (const int*)nullptr, 1, (const int*)nullptr,
                     (const double*)nullptr, 0);
  if (!hip_ok(ctx, hipGetLastError(), "scale_to_logvol launch")) return DH_ERR_HIP;
  if (!down(ctx, covs, d_c, (size_t)m * dd) || !down(ctx, ams, d_p, (size_t)m * dd) ||
      !down(ctx, axes, d_x, (size_t)m * dd) || !down(ctx, axlens, d_al, (size_t)m * d) ||
      !down(ctx, logvols, d_lv, (size_t)m))
    return DH_ERR_HIP;
  return dh_sync(ctx);
}

}  // extern "C"
