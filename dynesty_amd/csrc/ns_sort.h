// Register-resident bitonic sort of a run's live slots by (log-likelihood, slot) -- shared by ns.hip (ns_consume) and
// walkq.hip (itemgen_kernel's presort workgroups, round 6).  Test infrastructure: none.  Workgroups of 256 threads.
#pragma once
#include <hip/hip_runtime.h>

namespace dh_sort {
constexpr int kT = 256;

// the same without short-circuit evaluation (the compiler turns || and && on lane values into exec-mask branches:
// a dozen scalar branches per compare-exchange of the sort)
__device__ __forceinline__ int key_before_nb(double ka, int sa, double kb, int sb) {
  const int lt = ka < kb ? 1 : 0, eq = ka == kb ? 1 : 0, sl = sa < sb ? 1 : 0;
  return lt | (eq & sl);
}

// Bitonic sort of a run's slots by (log-likelihood, slot) with the elements in REGISTERS: thread t holds the SPT
// consecutive positions t * SPT .. of the P = SPT * kT, so a compare-exchange at distance jj is inside the thread
// (jj < SPT), a lane exchange inside the wavefront (jj < 64 SPT: __shfl_xor), and only the two or three widest
// distances go through LDS (the slots travel, the keys are looked up again).  The version that kept the order in
// LDS and read every key through its slot paid two dependent LDS round trips and a workgroup barrier for each of
// the 66 stages of 2048 elements: 55 us of the 150 us of a C2 queue consumption.
// sidx: max(P, kT) entries; padding = slot 0xFFFF / key +inf sorts to the end.
template <int SPT>
__device__ __attribute__((noinline)) void sort_slots(const double* skey, unsigned short* sidx, int N, int nsidx) {
  constexpr int P = SPT * kT;
  const int t = threadIdx.x, lane = t & 63, g0 = t * SPT;
  double k[SPT];
  int sl[SPT];
#pragma unroll
  for (int e = 0; e < SPT; ++e) {
    const int g = g0 + e;
    sl[e] = g < N ? g : 0xFFFF;
    k[e] = g < N ? skey[g] : INFINITY;
  }
  for (int kk = 2; kk <= P; kk <<= 1) {
    int jj = kk >> 1;
    // distances between wavefronts: through LDS
    for (; jj >= 64 * SPT; jj >>= 1) {
#pragma unroll
      for (int e = 0; e < SPT; ++e)
        if (g0 + e < nsidx) sidx[g0 + e] = (unsigned short)sl[e];
      __syncthreads();
      int ps[SPT];
      double pk[SPT];
#pragma unroll
      for (int e = 0; e < SPT; ++e) {
        const int gp = (g0 + e) ^ jj;
        ps[e] = gp < nsidx ? (int)sidx[gp] : 0xFFFF;
      }
#pragma unroll
      for (int e = 0; e < SPT; ++e) pk[e] = ps[e] == 0xFFFF ? INFINITY : skey[ps[e]];
#pragma unroll
      for (int e = 0; e < SPT; ++e) {
        const int g = g0 + e;
        const int keep_min = (((g & jj) == 0) == ((g & kk) == 0)) ? 1 : 0;
        // (a strict total order: "partner first" decides both directions; identical paddings swap harmlessly)
        const bool take = key_before_nb(pk[e], ps[e], k[e], sl[e]) == keep_min;
        k[e] = take ? pk[e] : k[e];
        sl[e] = take ? ps[e] : sl[e];
      }
      __syncthreads();
    }
    // distances between lanes
    for (; jj >= SPT; jj >>= 1) {
      const int m = jj / SPT;
      const bool lower = (lane & m) == 0;
      // (all exchanges issued before the first comparison: one LDS-crossbar latency per stage, not per element)
      int plo[SPT], phi[SPT], ps[SPT];
      const int src_lane = (lane ^ m) << 2;
#pragma unroll
      for (int e = 0; e < SPT; ++e) {
        const long long bits = __double_as_longlong(k[e]);
        plo[e] = __builtin_amdgcn_ds_bpermute(src_lane, (int)(unsigned)bits);
        phi[e] = __builtin_amdgcn_ds_bpermute(src_lane, (int)(unsigned)(bits >> 32));
        ps[e] = __builtin_amdgcn_ds_bpermute(src_lane, sl[e]);
      }
#pragma unroll
      for (int e = 0; e < SPT; ++e) {
        const double pk = __longlong_as_double(((long long)phi[e] << 32) | (unsigned)plo[e]);
        const int keep_min = (lower == (((g0 + e) & kk) == 0)) ? 1 : 0;
        const bool take = key_before_nb(pk, ps[e], k[e], sl[e]) == keep_min;
        k[e] = take ? pk : k[e];
        sl[e] = take ? ps[e] : sl[e];
      }
    }
    // distances inside the thread
#pragma unroll
    for (int J = SPT / 2; J >= 1; J >>= 1) {
      if (J <= (kk >> 1)) {
#pragma unroll
        for (int e = 0; e < SPT; ++e) {
          if ((e & J) == 0) {
            const int e2 = e | J;
            const int asc = (((g0 + e) & kk) == 0) ? 1 : 0;
            const bool sw = key_before_nb(k[e2], sl[e2], k[e], sl[e]) == asc;
            const double ka = k[e], kb = k[e2];
            const int sa = sl[e], sb = sl[e2];
            k[e] = sw ? kb : ka;
            k[e2] = sw ? ka : kb;
            sl[e] = sw ? sb : sa;
            sl[e2] = sw ? sa : sb;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < SPT; ++e)
    if (g0 + e < nsidx) sidx[g0 + e] = (unsigned short)sl[e];
  __syncthreads();
}

}  // namespace dh_sort
