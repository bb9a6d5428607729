// accuracy of v_rsq_f64 / v_rcp_f64 seeds and of k Newton steps (max relative error over a sweep of inputs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = 0.5 + 3.5 * (double)i / n;  // [0.5, 4)
  double y = __builtin_amdgcn_rsq(x);
  double ref = 1.0 / sqrt(x);
  for (int it = 0; it < 4; ++it) {
    out[(size_t)it * n + i] = fabs(y - ref) / ref;
    const double xy = x * y;
    const double e = fma(-xy, y, 1.0);
    y = fma(0.5 * y, e, y);
  }
  double r = __builtin_amdgcn_rcp(x);
  double rref = 1.0 / x;
  for (int it = 0; it < 4; ++it) {
    out[(size_t)(4 + it) * n + i] = fabs(r - rref) / rref;
    const double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
  }
}
int main() {
  const int n = 1 << 20;
  double* d;
  hipMalloc(&d, sizeof(double) * 8 * n);
  k<<<n / 256, 256>>>(d, n);
  double* h = new double[8 * (size_t)n];
  hipMemcpy(h, d, sizeof(double) * 8 * n, hipMemcpyDeviceToHost);
  for (int it = 0; it < 8; ++it) {
    double mx = 0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, h[(size_t)it * n + i]);
    printf("%s after %d Newton steps: max rel err %.3e (2^%.1f)\n", it < 4 ? "rsq" : "rcp", it & 3, mx, mx > 0 ? log2(mx) : -99.0);
  }
  return 0;
}
