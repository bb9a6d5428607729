// Calibration: sustained fp64 rate of (C) v_fma_f64 with VGPR operands,
// (S) with one SGPR operand, (M) v_mfma_f64_16x16x4_f64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) kC(double* out, int iters, double c) {
  double acc[24], m[8];
  for (int i = 0; i < 24; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  for (int i = 0; i < 8; ++i) m[i] = c + i * 1e-9 + threadIdx.x * 1e-12;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 24; ++i) acc[i] = fma(m[r], acc[i], 1e-9);
  }
  double s = 0;
  for (int i = 0; i < 24; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) kS(double* out, int iters, const double* cs) {
  double acc[24];
  for (int i = 0; i < 24; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double m[8];
  for (int i = 0; i < 8; ++i) m[i] = cs[i];   // uniform -> SGPR
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 24; ++i) acc[i] = fma(m[r], acc[i], 1e-9);
  }
  double s = 0;
  for (int i = 0; i < 24; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) kM(double* out, int iters, double c) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = c + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 2000;
  double *out, *cs; (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cs, 64);
  double h[8] = {0.999, 0.998, 0.997, 0.996, 0.995, 0.994, 0.993, 0.992};
  (void)hipMemcpy(cs, h, 64, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    float t;
    (void)hipEventRecord(e0); kC<<<blocks, 64>>>(out, iters, 0.999); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    printf("VGPR fma: %.3f ms %.2f TFLOP/s | ", t, 2.0 * blocks * 64 * iters * 192 / t / 1e9);
    (void)hipEventRecord(e0); kS<<<blocks, 64>>>(out, iters, cs); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    printf("SGPR fma: %.3f ms %.2f TFLOP/s | ", t, 2.0 * blocks * 64 * iters * 192 / t / 1e9);
    (void)hipEventRecord(e0); kM<<<blocks, 64>>>(out, iters, 0.5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    printf("MFMA f64 16x16x4: %.3f ms %.2f TFLOP/s\n", t, 2.0 * blocks * iters * 8 * 16 * 16 * 4 / t / 1e9);
  }
  return 0;
}
