// Micro-benchmark: 25x25 fp64 mat-vec per lane, matrix wave-uniform.
//  A: matrix rows through the scalar cache (SGPR operands)
//  B: matrix rows in a VGPR pair, element picked by DPP row_newbcast
// build: hipcc -O3 --offload-arch=gfx950 dpp_matvec.hip -o dpp_matvec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef const __attribute__((address_space(4))) double* cdptr;
constexpr int N = 25, NP = 32;

template <int I>
__device__ __forceinline__ void fmac_bcast(double& acc, double row, double d) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(row), "v"(d), "n"(I));
}
template <int I>
struct Sweep {
  static __device__ __forceinline__ void run(double (&acc)[N], double r0, double r1, double d) {
    fmac_bcast<(I & 15)>(acc[I], I < 16 ? r0 : r1, d);
    if constexpr (I + 1 < N) Sweep<I + 1>::run(acc, r0, r1, d);
  }
};

__global__ void __launch_bounds__(64) kB(const double* __restrict__ MT, const double* __restrict__ x,
                                         double* out, int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  const int l16 = lane & 15;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double r0 = MT[j * NP + l16], r1 = MT[j * NP + 16 + l16];
      double d = xs[j * 64 + lane];
      Sweep<0>::run(acc, r0, r1, d);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;  // keep the loop live
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

__global__ void __launch_bounds__(64) kA(const double* MT, const double* __restrict__ x, double* out,
                                         int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  cdptr M = (cdptr)(unsigned long long)MT;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      double d = xs[j * 64 + lane];
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = fma(M[j * NP + i], d, acc[i]);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// A5: as A, the 25 input components fetched five at a time (five LDS reads in flight)
__global__ void __launch_bounds__(64) kA5(const double* MT, const double* __restrict__ x, double* out,
                                          int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  cdptr M = (cdptr)(unsigned long long)MT;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int j0 = 0; j0 < N; j0 += 5) {
      double d[5];
#pragma unroll
      for (int u = 0; u < 5; ++u) d[u] = xs[(j0 + u) * 64 + lane];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = fma(M[(j0 + u) * NP + i], d[u], acc[i]);
      }
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

__global__ void __launch_bounds__(64) kL(const double* __restrict__ MT, const double* __restrict__ x,
                                         double* out, int iters) {
  __shared__ double xs[N * 64];
  __shared__ __attribute__((aligned(16))) double ms[N * NP];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  for (int e = lane; e < N * NP; e += 64) ms[e] = MT[e];
  __syncthreads();
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 5
    for (int j = 0; j < N; ++j) {
      double d = xs[j * 64 + lane];
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = fma(ms[j * NP + i], d, acc[i]);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// explicit double-buffered scalar loads (inline asm: the compiler cannot sink them)
typedef double d8v __attribute__((ext_vector_type(8)));
typedef double d4v __attribute__((ext_vector_type(4)));
struct Half { d8v a; d4v b; double c; };   // 13 doubles = 26 SGPRs
__device__ __forceinline__ void load_half(Half& h, const double* p) {
  asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx2 %2, %3, 0x60"
               : "=&s"(h.a), "=&s"(h.b), "=&s"(h.c) : "s"(p));
}
__device__ __forceinline__ void wait_half(Half& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(h.a), "+s"(h.b), "+s"(h.c));
}
__device__ __forceinline__ void fma_half(const Half& h, double d, double* acc) {
  acc[0] = fma(h.a[0], d, acc[0]); acc[1] = fma(h.a[1], d, acc[1]); acc[2] = fma(h.a[2], d, acc[2]);
  acc[3] = fma(h.a[3], d, acc[3]); acc[4] = fma(h.a[4], d, acc[4]); acc[5] = fma(h.a[5], d, acc[5]);
  acc[6] = fma(h.a[6], d, acc[6]); acc[7] = fma(h.a[7], d, acc[7]); acc[8] = fma(h.b[0], d, acc[8]);
  acc[9] = fma(h.b[1], d, acc[9]); acc[10] = fma(h.b[2], d, acc[10]); acc[11] = fma(h.b[3], d, acc[11]);
  acc[12] = fma(h.c, d, acc[12]);
}
__global__ void __launch_bounds__(64) kP(const double* MT, const double* __restrict__ x, double* out,
                                         int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  __syncthreads();
  double acc[26];
#pragma unroll
  for (int i = 0; i < 26; ++i) acc[i] = 0;
  for (int it = 0; it < iters; ++it) {
    Half h0, h1;
    load_half(h0, MT);
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      const double d = xs[j * 64 + lane];
      wait_half(h0);                             // h0 landed; nothing else outstanding
      load_half(h1, MT + j * NP + 13);           // flies during the 13 FMAs below
      fma_half(h0, d, acc);
      const int jn = j + 1 < N ? j + 1 : j;
      wait_half(h1);
      load_half(h0, MT + jn * NP);               // flies during the next 13 FMAs
      fma_half(h1, d, acc + 13);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// one asm block = {wait for `cur`; issue the loads of `nxt`; 13 FMAs from `cur`}: the
// compiler cannot separate the loads from the math they are meant to overlap with.
__device__ __forceinline__ void step_half(const Half& cur, Half& nxt, const double* pn, double d,
                                          double* acc) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_load_dwordx16 %13, %17, 0x0\n\t"
      "s_load_dwordx8 %14, %17, 0x40\n\t"
      "s_load_dwordx2 %15, %17, 0x60\n\t"
      "v_fmac_f64 %0, %18, %16 \n\t"
      "v_fmac_f64 %1, %19, %16 \n\t"
      "v_fmac_f64 %2, %20, %16 \n\t"
      "v_fmac_f64 %3, %21, %16 \n\t"
      "v_fmac_f64 %4, %22, %16 \n\t"
      "v_fmac_f64 %5, %23, %16 \n\t"
      "v_fmac_f64 %6, %24, %16 \n\t"
      "v_fmac_f64 %7, %25, %16 \n\t"
      "v_fmac_f64 %8, %26, %16 \n\t"
      "v_fmac_f64 %9, %27, %16 \n\t"
      "v_fmac_f64 %10, %28, %16 \n\t"
      "v_fmac_f64 %11, %29, %16 \n\t"
      "v_fmac_f64 %12, %30, %16"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
        "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
        "=&s"(nxt.a), "=&s"(nxt.b), "=&s"(nxt.c)
      : "v"(d), "s"(pn), "s"(cur.a[0]), "s"(cur.a[1]), "s"(cur.a[2]), "s"(cur.a[3]), "s"(cur.a[4]),
        "s"(cur.a[5]), "s"(cur.a[6]), "s"(cur.a[7]), "s"(cur.b[0]), "s"(cur.b[1]), "s"(cur.b[2]),
        "s"(cur.b[3]), "s"(cur.c));
}
__global__ void __launch_bounds__(64) kQ(const double* MT, const double* __restrict__ x, double* out,
                                         int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  __syncthreads();
  double acc[26];
#pragma unroll
  for (int i = 0; i < 26; ++i) acc[i] = 0;
  for (int it = 0; it < iters; ++it) {
    Half h0, h1;
    load_half(h0, MT);
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      const double d = xs[j * 64 + lane];
      const int jn = j + 1 < N ? j + 1 : j;
      step_half(h0, h1, MT + j * NP + 13, d, acc);       // h1 flies during the first 13 FMAs
      step_half(h1, h0, MT + jn * NP, d, acc + 13);      // next row's h0 flies during these
    }
    wait_half(h0);
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// M: the mat-vec of all 64 lanes as ONE GEMM on the fp64 matrix cores: C[i][n] = sum_k A[i][k] B[k][n],
// A = the (wave-uniform) matrix held in registers for the whole kernel (no scalar loads per step),
// B[k][n] = component k of lane n's vector read from the [dim][lane] LDS columns, C transposed back to
// "lane n holds all components" through LDS.
typedef double mfma4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) kM(const double* __restrict__ MT, const double* __restrict__ x,
                                         double* out, int iters) {
  __shared__ double xs[N * 64];
  __shared__ double ys[NP * 64];
  const int lane = threadIdx.x;
  const int lj = lane & 15, lk = lane >> 4;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  constexpr int KS = (N + 3) / 4;  // 7
  double afr[2][KS];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 4 + lk;
      afr[mb][ks] = k < N ? MT[k * NP + mb * 16 + lj] : 0.0;
    }
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  for (int it = 0; it < iters; ++it) {
    mfma4 c[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) c[mb][nb] = (mfma4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 4 + lk;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const double b = k < N ? xs[k * 64 + nb * 16 + lj] : 0.0;
        c[0][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[0][ks], b, c[0][nb], 0, 0, 0);
        c[1][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[1][ks], b, c[1][nb], 0, 0, 0);
      }
    }
    // C[row = lk + 4 r (+16 mb)][col = lj (+16 nb)] -> ys[row][lane = col]
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ys[(mb * 16 + lk + 4 * r) * 64 + nb * 16 + lj] = c[mb][nb][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] += ys[i * 64 + lane];
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// H: hybrid operand feed -- of each matrix row (one input component j, 25 outputs) the first LH outputs come
// through LDS broadcast reads, the rest through the scalar cache: two independent per-CU data paths.
template <int LH>
__global__ void __launch_bounds__(64) kH(const double* MT, const double* __restrict__ x, double* out, int iters) {
  __shared__ double xs[N * 64];
  __shared__ __attribute__((aligned(16))) double ms[N * NP];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  for (int e = lane; e < N * NP; e += 64) ms[e] = MT[e];
  __syncthreads();
  double acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0;
  cdptr M = (cdptr)(unsigned long long)MT;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      double d = xs[j * 64 + lane];
#pragma unroll
      for (int i = 0; i < LH; ++i) acc[i] = fma(ms[j * NP + i], d, acc[i]);
#pragma unroll
      for (int i = LH; i < N; ++i) acc[i] = fma(M[j * NP + i], d, acc[i]);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// R: ceiling of the loop shape -- the same 25 FMAs per input component, matrix row held in VGPRs / SGPRs for the
// whole kernel (wrong mathematics, right instruction mix): what the mat-vec would do with free operand delivery.
template <bool SG, int UN>
__global__ void __launch_bounds__(64) kR(const double* MT, const double* __restrict__ x, double* out, int iters) {
  __shared__ double xs[N * 64];
  const int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) xs[j * 64 + lane] = x[(size_t)blockIdx.x * N * 64 + j * 64 + lane];
  double acc[N], row[N];
  cdptr M = (cdptr)(unsigned long long)MT;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    acc[i] = 0;
    row[i] = SG ? M[i] : MT[i + (lane & 1)];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll UN
    for (int j = 0; j < N; ++j) {
      double d = xs[j * 64 + lane];
#pragma unroll
      for (int i = 0; i < N; ++i) acc[i] = fma(row[i], d, acc[i]);
    }
    xs[(it % N) * 64 + lane] = acc[it % N] * 1e-3;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[(size_t)blockIdx.x * N * 64 + i * 64 + lane] = acc[i];
}

// A2: as A, but every lane carries TWO vectors (two walkers per lane): each scalar operand feeds two
// FMAs, so the scalar data path carries half the dwords per flop
template <int W>
__global__ void __launch_bounds__(64) kAW(const double* MT, const double* __restrict__ x, double* out,
                                         int iters) {
  __shared__ double xs[W * N * 64];
  const int lane = threadIdx.x;
  for (int w = 0; w < W; ++w)
    for (int j = 0; j < N; ++j)
      xs[(w * N + j) * 64 + lane] = x[((size_t)blockIdx.x * W + w) * N * 64 + j * 64 + lane];
  double acc[W][N];
#pragma unroll
  for (int w = 0; w < W; ++w)
#pragma unroll
    for (int i = 0; i < N; ++i) acc[w][i] = 0;
  cdptr M = (cdptr)(unsigned long long)MT;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
      double d[W];
#pragma unroll
      for (int w = 0; w < W; ++w) d[w] = xs[(w * N + j) * 64 + lane];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const double m = M[j * NP + i];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w][i] = fma(m, d[w], acc[w][i]);
      }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) xs[(w * N + it % N) * 64 + lane] = acc[w][it % N] * 1e-3;
  }
#pragma unroll
  for (int w = 0; w < W; ++w)
#pragma unroll
    for (int i = 0; i < N; ++i) out[((size_t)blockIdx.x * W + w) * N * 64 + i * 64 + lane] = acc[w][i];
}

int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 90;
  std::vector<double> MT(N * NP, 0.0), x((size_t)blocks * N * 64);
  srand(1);
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) MT[j * NP + i] = (rand() / (double)RAND_MAX - 0.5) * 0.3;
  for (auto& v : x) v = rand() / (double)RAND_MAX - 0.5;
  double *dM, *dx, *dA, *dB;
  hipMalloc(&dM, MT.size() * 8); hipMalloc(&dx, x.size() * 8);
  hipMalloc(&dA, x.size() * 8); hipMalloc(&dB, x.size() * 8);
  hipMemcpy(dM, MT.data(), MT.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    float ma, mb;
    hipEventRecord(e0); kA<<<blocks, 64>>>(dM, dx, dA, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ma, e0, e1);
    hipEventRecord(e0); kB<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&mb, e0, e1);
    float mp;
    hipEventRecord(e0); kQ<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&mp, e0, e1);
    printf("asm-block: %.3f ms %.2f TFLOP/s | ", mp, (double)blocks * 64 * iters * N * N * 2 / mp / 1e9);
    float ml;
    hipEventRecord(e0); kL<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ml, e0, e1);
    printf("lds-bcast: %.3f ms %.2f TFLOP/s | ", ml, (double)blocks * 64 * iters * N * N * 2 / ml / 1e9);
    hipEventRecord(e0); kB<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    {
      float mh;
      hipEventRecord(e0); kH<8><<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&mh, e0, e1);
      printf("hybrid8: %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / mh / 1e9);
      hipEventRecord(e0); kH<12><<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&mh, e0, e1);
      printf("hybrid12: %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / mh / 1e9);
    }
    {
      float mr;
      hipEventRecord(e0); kR<false, 1><<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&mr, e0, e1);
      printf("ceiling(vgpr row): %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / mr / 1e9);
      hipEventRecord(e0); kR<true, 5><<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&mr, e0, e1);
      printf("ceiling(sgpr row, unroll 5): %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / mr / 1e9);
    }
    {
      float m5;
      hipEventRecord(e0); kA5<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&m5, e0, e1);
      printf("sgpr x5: %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / m5 / 1e9);
    }
    {
      float m2;
      hipEventRecord(e0); kAW<2><<<blocks / 2, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&m2, e0, e1);
      printf("sgpr W=2: %.2f TF | ", (double)blocks * 64 * iters * N * N * 2 / m2 / 1e9);
      hipEventRecord(e0); kAW<3><<<blocks / 3, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&m2, e0, e1);
      printf("sgpr W=3: %.2f TF | ", (double)(blocks / 3) * 3 * 64 * iters * N * N * 2 / m2 / 1e9);
    }
    float mm;
    hipEventRecord(e0); kM<<<blocks, 64>>>(dM, dx, dB, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&mm, e0, e1);
    printf("mfma: %.3f ms %.2f TFLOP/s | ", mm, (double)blocks * 64 * iters * N * N * 2 / mm / 1e9);
    double fl = (double)blocks * 64 * iters * N * N * 2;
    printf("sgpr: %.3f ms %.2f TFLOP/s | dpp: %.3f ms %.2f TFLOP/s\n", ma, fl / ma / 1e9, mb, fl / mb / 1e9);
  }
  kA<<<blocks, 64>>>(dM, dx, dA, iters);
  kM<<<blocks, 64>>>(dM, dx, dB, iters);
  hipDeviceSynchronize();
  std::vector<double> a(x.size()), b(x.size());
  hipMemcpy(a.data(), dA, a.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), dB, b.size() * 8, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs(a[i] - b[i])); mx = fmax(mx, fabs(a[i])); }
  printf("max |sgpr - mfma| = %.3e (max |val| %.3e) -> %s\n", md, mx, md <= 1e-12 * fmax(1.0, mx) ? "MATCH" : "MISMATCH");
  return 0;
}
