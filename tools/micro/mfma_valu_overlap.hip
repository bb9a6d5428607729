// Do f64 matrix instructions and f64 vector instructions of a SIMD overlap on the MI355X?  One kernel issues NM
// v_mfma_f64_16x16x4 per loop trip (four independent accumulator chains), one NV v_fma_f64 (eight independent chains),
// one both -- in one wavefront's instruction stream (interleaved by hand) or from different wavefronts of the SIMD
// (even wavefronts matrix, odd wavefronts vector) -- at W wavefronts per SIMD.  (round 6: rwalkq_kernel's step is 28
// matrix instructions + ~320 vector instructions and its time did not move when the dependent chain was cut)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: matrix only, 1: vector only, 2: both in one stream, 3: even waves matrix / odd waves vector
__global__ void __launch_bounds__(256) k(double* out, int trips, double seed) {
  const int wave = threadIdx.x >> 6;
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = seed + threadIdx.x, y = seed * 0.5;
  double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int i = 0; i < trips; ++i) {
    if (do_m && do_v) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
        f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
        f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
        f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
        f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
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
        f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
        f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int MODE>
float run(double* out, int blocks, int trips) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, trips, 1e-30);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, trips, 1e-30);
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
    printf("%d wavefront(s) per SIMD: matrix only %8.1f us (%5.1f cycles/instr @2.4GHz) | vector only %8.1f us (%4.1f cycles/instr) | "
           "both, one stream %8.1f us (sum %8.1f) | even waves matrix, odd waves vector %8.1f us\n",
           wps, tm, tm * 2400.0 / (trips * 8.0 * wps), tv, tv * 2400.0 / (trips * 64.0 * wps), tb, tm + tv, ts);
  }
  return 0;
}
