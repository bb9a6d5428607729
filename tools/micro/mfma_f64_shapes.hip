// Layout and cost of the two fp64 matrix instructions on gfx950:
//   v_mfma_f64_16x16x4_f64      (one 16x16x4 product, 4 accumulator values per lane)
//   v_mfma_f64_4x4x4_4b_f64     (four independent 4x4x4 products, 1 value per lane)
// Prints D of the 4x4x4 form for A = 1 + lane, B = 100 + lane (the host script checks which
// (block, i, k) / (block, k, j) / (block, i, j) lane maps reproduce it) and the cycles per
// instruction of both forms in dependent and independent chains.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_shapes.hip -o /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(double* out) {
  const int l = threadIdx.x;
  const double a = 1.0 + l, b = 100.0 + l;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[l] = d;
}

__global__ void time_kernel(long long* out, int n) {
  const int l = threadIdx.x;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  // 16x16x4, two alternating accumulators
  v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
  }
  long long t1 = clock64();
  // 4x4x4_4b, two alternating accumulators
  double d0 = 0, d1 = 0;
  for (int i = 0; i < n; ++i) {
    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, d1, 0, 0, 0);
  }
  long long t2 = clock64();
  // 4x4x4_4b, eight independent accumulators
  double e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a + q, b, e[q], 0, 0, 0);
  }
  long long t3 = clock64();
  // plain FMA chain for scale: 8 independent
  double f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = fma(f[q], a, b);
  }
  long long t4 = clock64();
  double sink = c0[0] + c1[1] + d0 + d1;
  for (int q = 0; q < 8; ++q) sink += e[q] + f[q];
  if (l == 0) {
    out[0] = t1 - t0;
    out[1] = t2 - t1;
    out[2] = t3 - t2;
    out[3] = t4 - t3;
    out[4] = (long long)sink;
  }
}

int main() {
  double* d;
  long long* t;
  hipMalloc(&d, 64 * 8);
  hipMalloc(&t, 64);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d);
  double h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("D:");
  for (int i = 0; i < 64; ++i) printf(" %.0f", h[i]);
  printf("\n");
  const int n = 2000;
  hipLaunchKernelGGL(time_kernel, dim3(1), dim3(64), 0, 0, t, n);
  long long ht[8];
  hipMemcpy(ht, t, 40, hipMemcpyDeviceToHost);
  printf("cycles per instruction: 16x16x4 (2 chains) %.1f | 4x4x4_4b (2 chains) %.1f | 4x4x4_4b (8 chains) %.1f | v_fma_f64 (8 chains) %.1f\n",
         ht[0] / (2.0 * n), ht[1] / (2.0 * n), ht[2] / (8.0 * n), ht[3] / (8.0 * n));
  return 0;
}
