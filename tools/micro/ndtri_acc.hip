// accuracy of ndtri variants on the device: writes p, x_plain = -sqrt2 erfcinv(2p), x_newton (+1 Newton step on erfc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* p, double* a, double* b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = -1.4142135623730951 * erfcinv(2.0 * p[i]);
  a[i] = x;
  double f = 0.5 * erfc(-x * 0.7071067811865476) - p[i];
  double pdf = 0.3989422804014327 * exp(-0.5 * x * x);
  if (pdf > 1e-300) x -= f / pdf;
  b[i] = x;
}
int main() {
  const int n = 1 << 20;
  double* hp = new double[n];
  for (int i = 0; i < n; ++i) {
    double u = (i + 0.5) / n;
    hp[i] = (i % 3 == 0) ? u : (i % 3 == 1 ? pow(10.0, -300.0 * u) : 1.0 - pow(10.0, -15.0 * u));
  }
  double *dp, *da, *db;
  (void)hipMalloc(&dp, n * 8); (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8);
  (void)hipMemcpy(dp, hp, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dp, da, db, n);
  double* ha = new double[n]; double* hb = new double[n];
  (void)hipMemcpy(ha, da, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, db, n * 8, hipMemcpyDeviceToHost);
  FILE* f = fopen("gpurun_out/ndtri_acc.bin", "wb");
  fwrite(hp, 8, n, f); fwrite(ha, 8, n, f); fwrite(hb, 8, n, f); fclose(f);
  printf("wrote %d\n", n);
  return 0;
}
