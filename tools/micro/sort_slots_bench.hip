// Times ns.hip's register bitonic sort (sort_slots<8>, 2048 slots) in isolation: clock64 cycles and wall time.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I dynesty_amd/csrc tools/micro/sort_slots_bench.hip -o tools/micro/sort_slots_bench
#include "../../dynesty_amd/csrc/ns.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <chrono>

__global__ void __launch_bounds__(kT) bench_kernel(const double* keys, int N, unsigned short* out, long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* skey = (double*)smem;
  unsigned short* sidx = (unsigned short*)(skey + N);
  for (int i = threadIdx.x; i < N; i += kT) skey[i] = keys[(size_t)blockIdx.x * N + i];
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) sort_slots<8>(skey, sidx, N, 2048);
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  for (int i = threadIdx.x; i < 2048; i += kT) out[(size_t)blockIdx.x * 2048 + i] = sidx[i];
}

int main() {
  const int N = 2000, B = 64, reps = 10;
  std::vector<double> h((size_t)B * N);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0; }
  double* d; unsigned short* o; long long* c;
  hipMalloc(&d, h.size() * 8); hipMalloc(&o, (size_t)B * 2048 * 2); hipMalloc(&c, B * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = (size_t)N * 8 + 2048 * 2 + 64;
  hipFuncSetAttribute((const void*)bench_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int it = 0; it < 3; ++it) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(bench_kernel, dim3(B), dim3(kT), lds, 0, d, N, o, c, reps);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    long long hc[B];
    hipMemcpy(hc, c, sizeof hc, hipMemcpyDeviceToHost);
    printf("%d sorts of %d: %.1f us per sort (wall incl. launch), %lld cycles per sort\n", reps, N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, hc[0] / reps);
  }
  std::vector<unsigned short> ho((size_t)B * 2048);
  hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i + 1 < N; ++i) {
      const double ka = h[(size_t)b * N + ho[(size_t)b * 2048 + i]], kb = h[(size_t)b * N + ho[(size_t)b * 2048 + i + 1]];
      if (ka > kb) ++bad;
    }
  printf("order violations: %d\n", bad);
  return bad != 0;
}
