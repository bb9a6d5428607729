// How fast does the MI355X hand out workgroups?  An empty kernel (one global load so that it is not optimised away) over N
// workgroups of T threads with S bytes of dynamic LDS: time per launch by HIP events.  (round 6: the level kernels of the
// rebuild launch 1 000 - 2 700 workgroups each)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_rate.hip -o tools/micro/dispatch_rate && tools/micro/dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty(const int* p, int* out) {
  extern __shared__ int smem[];
  if (p[blockIdx.x & 1023] == 12345) out[0] = smem[threadIdx.x];
}
__global__ void k_busy(const int* p, int* out, int spin) {  // ~spin x 64 cycles of work per workgroup
  extern __shared__ int smem[];
  int v = p[blockIdx.x & 1023];
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(1);
  if (v == 12345) out[0] = smem[threadIdx.x];
}
int main() {
  int *p, *out;
  hipMalloc(&p, 4096); hipMemset(p, 0, 4096); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int Ts[] = {64, 256, 1024};
  const int Ss[] = {0, 31 * 1024, 77 * 1024};
  const int Ns[] = {256, 1024, 2048, 4096, 16384};
  for (int S : Ss) {
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, S);
    hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, S);
    for (int T : Ts)
      for (int N : Ns) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty, dim3(N), dim3(T), S, 0, p, out);
        hipDeviceSynchronize();
        const int reps = 20;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_empty, dim3(N), dim3(T), S, 0, p, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_busy, dim3(N), dim3(T), S, 0, p, out, 500);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("LDS %6d B  threads %4d  workgroups %6d : empty %8.2f us per launch (%6.1f wg/us)   busy(~15us each) %8.2f us\n", S, T, N,
               ms / reps * 1e3, N / (ms / reps * 1e3), ms2 / reps * 1e3);
      }
  }
  return 0;
}
