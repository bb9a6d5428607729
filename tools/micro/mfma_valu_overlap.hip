// Do f64 matrix instructions and f64 vector instructions of a SIMD overlap on the MI355X?  One kernel issues NM
// v_mfma_f64_16x16x4 per loop trip (four independent accumulator chains), one NV v_fma_f64 (eight independent chains),
// one both -- in one wavefront's instruction stream (interleaved by hand) or from different wavefronts of the SIMD
// (even wavefronts matrix, odd wavefronts vector) -- at W wavefronts per SIMD.  (round 6: rwalkq_kernel's step is 28
// matrix instructions + ~320 vector instructions and its time did not move when the dependent chain was cut)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
// INT: the vector work is integer (v_mad_u64_u32 chains, itemgen_kernel's instruction) instead of v_fma_f64
template <int MODE, bool INT = false>  // 0: matrix only, 1: vector only, 2: both in one stream, 3: even waves matrix / odd waves vector
__global__ void __launch_bounds__(256) k(double* out, int trips, double seed) {
  const int wave = threadIdx.x >> 6;
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = seed + threadIdx.x, y = seed * 0.5;
  double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
  unsigned long long u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  const unsigned mlt = 0x9E3779B9u + threadIdx.x;
#define IMAD(v) v = (unsigned long long)(unsigned)v * mlt + v
#define VEC8() do { if (INT) { IMAD(u0); IMAD(u1); IMAD(u2); IMAD(u3); IMAD(u4); IMAD(u5); IMAD(u6); IMAD(u7); } else { f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x); } } while (0)
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int i = 0; i < trips; ++i) {
    if (do_m && do_v) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        VEC8();
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        VEC8();
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        VEC8();
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        VEC8();
      }
    } else if (do_m) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
      }
    } else if (do_v) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        VEC8();
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (double)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}
template <int MODE, bool INT = false>
float run(double* out, int blocks, int trips) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, INT>), dim3(blocks), dim3(256), 0, 0, out, trips, 1e-30);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, INT>), dim3(blocks), dim3(256), 0, 0, out, trips, 1e-30);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main() {
  double* out; hipMalloc(&out, 8 * 256 * 4096);
  const int trips = 2000;  // per trip and wavefront: 8 matrix instructions and / or 64 vector FMAs
  for (int wps : {1, 2, 4}) {  // wavefronts per SIMD: blocks of 4 waves, `wps` blocks per CU
    const int blocks = 256 * wps;
    const float tm = run<0>(out, blocks, trips), tv = run<1>(out, blocks, trips), tb = run<2>(out, blocks, trips),
                ts = run<3>(out, blocks, trips);
    const float iv = run<1, true>(out, blocks, trips), ib = run<2, true>(out, blocks, trips), is = run<3, true>(out, blocks, trips);
    printf("%d wavefront(s) per SIMD, INTEGER vector work (v_mad_u64_u32): vector only %8.1f us (%4.1f cycles/instr) | both, one stream %8.1f us (sum %8.1f) | "
           "even waves matrix, odd waves integer %8.1f us\n", wps, iv, iv * 2400.0 / (trips * 64.0 * wps), ib, tm + iv, is);
    printf("%d wavefront(s) per SIMD: matrix only %8.1f us (%5.1f cycles/instr @2.4GHz) | vector only %8.1f us (%4.1f cycles/instr) | "
           "both, one stream %8.1f us (sum %8.1f) | even waves matrix, odd waves vector %8.1f us\n",
           wps, tm, tm * 2400.0 / (trips * 8.0 * wps), tv, tv * 2400.0 / (trips * 64.0 * wps), tb, tm + tv, ts);
  }
  return 0;
}
