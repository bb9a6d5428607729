// Issue cost (cycles per wave64 instruction per SIMD) of the vector instructions the PCG64 round of walkq.hip is
// made of, measured with W wavefronts per SIMD all running the same loop of independent instructions:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_costs.hip -o tools/micro/valu_costs && tools/micro/valu_costs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define BODY(name, decl, stmt, fin)                                                     \
  __global__ void __launch_bounds__(256) k_##name(unsigned long long* out, int iters) {   \
    decl;                                                                               \
    const long long t0 = clock64();                                                     \
    for (int it = 0; it < iters; ++it) { REP8(REP8(stmt)) }                             \
    const long long t1 = clock64();                                                     \
    fin;                                                                                \
    if (threadIdx.x == 0) out[2 * blockIdx.x + 1] = (unsigned long long)(t1 - t0);      \
  }

// 64 instructions per iteration, 8 independent chains
#define U32D unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 1
#define U32F out[2 * blockIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define U64D unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned b = blockIdx.x | 3
#define U64F out[2 * blockIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define F64D unsigned tid = threadIdx.x; double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0 + 1e-9 * blockIdx.x
#define F64F out[2 * blockIdx.x] = (unsigned long long)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define A8(op) op(a0) op(a1) op(a2) op(a3) op(a4) op(a5) op(a6) op(a7)

#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(b) : "vcc");
#define OP_LSHL64(x) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(x));
#define OP_LSHLV64(x) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_ADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(x));
#define OP_CMP64(x) asm volatile("v_cmp_lt_u64 vcc, %0, %0" : : "v"(x) : "vcc");
#define OP_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define OP_ADDF64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_MULF64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_CVT(V_) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(V_) : "v"(tid));
#define OP_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
#define OP_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define OP_EXPF(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define OP_SUBCO(x) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(b) : "vcc");
#define OP_READLANE(x) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x) : "s20");
#define OP_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_SADD(x) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define OP_SLSHL64(x) asm volatile("s_lshl_b64 s[20:21], s[20:21], 1" : : : "s20", "s21", "scc");
#define OP_SFF1(x) asm volatile("s_ff1_i32_b64 s22, s[20:21]" : : : "s22");

BODY(add_u32, U32D, A8(OP_ADD), U32F)
BODY(mul_lo_u32, U32D, A8(OP_MULLO), U32F)
BODY(mul_hi_u32, U32D, A8(OP_MULHI), U32F)
BODY(mul_u32_u24, U32D, A8(OP_MUL24), U32F)
BODY(alignbit, U32D, A8(OP_ALIGN), U32F)
BODY(cndmask, U32D, A8(OP_CNDMASK), U32F)
BODY(sub_co, U32D, A8(OP_SUBCO), U32F)
BODY(exp_f32, U32D, A8(OP_EXPF), U32F)
BODY(readlane, U32D, A8(OP_READLANE), U32F)
BODY(mbcnt, U32D, A8(OP_MBCNT), U32F)
BODY(mad_u64_u32, U64D, A8(OP_MAD64), U64F)
BODY(lshl_b64_const, U64D, A8(OP_LSHL64), U64F)
BODY(lshl_b64_var, U64D, A8(OP_LSHLV64), U64F)
BODY(lshl_add_u64, U64D, A8(OP_ADD64), U64F)
BODY(cmp_lt_u64, U64D, A8(OP_CMP64), U64F)
BODY(fma_f64, F64D, A8(OP_FMA64), F64F)
BODY(add_f64, F64D, A8(OP_ADDF64), F64F)
BODY(mul_f64, F64D, A8(OP_MULF64), F64F)
BODY(cvt_f64_u32, F64D, A8(OP_CVT), F64F)
BODY(ldexp_f64, F64D, A8(OP_LDEXP), F64F)
BODY(s_add_u32, U32D, A8(OP_SADD), U32F)
BODY(s_lshl_b64, U32D, A8(OP_SLSHL64), U32F)
BODY(s_ff1_b64, U32D, A8(OP_SFF1), U32F)

typedef void (*kern_t)(unsigned long long*, int);
struct Entry { const char* name; kern_t k; };
#define E(n) {#n, k_##n}
int main(int argc, char** argv) {
  const int iters = 200;
  Entry es[] = {E(add_u32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(alignbit), E(cndmask), E(sub_co), E(exp_f32),
                E(readlane), E(mbcnt), E(mad_u64_u32), E(lshl_b64_const), E(lshl_b64_var), E(lshl_add_u64), E(cmp_lt_u64),
                E(fma_f64), E(add_f64), E(mul_f64), E(cvt_f64_u32), E(ldexp_f64), E(s_add_u32), E(s_lshl_b64), E(s_ff1_b64)};
  int dev_cus = 256;
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0); dev_cus = pr.multiProcessorCount;
  unsigned long long* out; (void)hipMalloc(&out, 16 * 4096);
  unsigned long long h[8192];
  printf("{\"cus\": %d, \"unit\": \"cycles per wave64 instruction, per wavefront (s_memtime), W wavefronts per SIMD\",\n \"ops\": {\n", dev_cus);
  const int n = sizeof(es) / sizeof(es[0]);
  for (int i = 0; i < n; ++i) {
    printf("  \"%s\": {", es[i].name);
    for (int w = 1; w <= 4; w *= 2) {  // W wavefronts per SIMD: w blocks of 256 threads per CU
      const int blocks = dev_cus * w;
      hipLaunchKernelGGL(es[i].k, dim3(blocks), dim3(256), 0, 0, out, 10);
      hipLaunchKernelGGL(es[i].k, dim3(blocks), dim3(256), 0, 0, out, iters);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, out, 16 * blocks, hipMemcpyDeviceToHost);
      double s = 0;
      for (int b = 0; b < blocks; ++b) s += (double)h[2 * b + 1];
      printf("\"W%d\": %.2f%s", w, s / blocks / (iters * 64.0), w < 4 ? ", " : "");
    }
    printf("}%s\n", i + 1 < n ? "," : "");
  }
  printf(" }\n}\n");
  return 0;
}
