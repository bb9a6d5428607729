// Device random streams that are bit-compatible with the generator dynesty
// hands to every proposal: numpy.random.Generator(numpy.random.PCG64(seed))
// (reference utils.py:993-1009 get_random_generator / get_seed_sequence).
//
//  * SeedSequence child hashing      numpy/random/bit_generator.pyx (SeedSequence)
//  * PCG64 = pcg_setseq_128 + XSL-RR numpy/random/src/pcg64/pcg64.h
//  * standard_normal                 256-layer ziggurat, numpy distributions.c
//  * random()                        (u64 >> 11) * 2^-53
//  * shuffle / bounded ints          masked rejection on buffered 32-bit halves
//
// Restated from the published algorithms; validated bit-for-bit against NumPy
// (tests/test_rng_host.py on CPU for the restatement, tests/test_gpu_rng.py on
// the device).  One generator per walker, held in registers: 128-bit state,
// 128-bit increment, plus NumPy's buffered uint32 half.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dh {

struct U128 {
  uint64_t hi, lo;
};

__host__ __device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// low 128 bits of a*b
__host__ __device__ __forceinline__ U128 mul128(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo * b.lo;
  r.hi = mulhi64(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
  return r;
}

__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}

#define DH_PCG_MULT_HI 0x2360ED051FC65DA4ull
#define DH_PCG_MULT_LO 0x4385DF649FCCF645ull

// SeedSequence constants (bit_generator.pyx)
#define DH_SS_INIT_A 0x43b0d7e5u
#define DH_SS_MULT_A 0x931e8875u
#define DH_SS_INIT_B 0x8b51f9ddu
#define DH_SS_MULT_B 0x58f38dedu
#define DH_SS_MIX_L 0xca01f9ddu
#define DH_SS_MIX_R 0x4973f715u

struct Pcg64 {
  U128 state;
  U128 inc;
  uint32_t has32;  // NumPy's buffered upper half (bitgen next_uint32)
  uint32_t buf32;

  __host__ __device__ __forceinline__ void step() {
    U128 m = {DH_PCG_MULT_HI, DH_PCG_MULT_LO};
    state = add128(mul128(state, m), inc);
  }
  // pcg64_next64: step, then XSL-RR of the NEW state
  __host__ __device__ __forceinline__ uint64_t next64() {
    step();
    uint64_t x = state.hi ^ state.lo;
    unsigned r = (unsigned)(state.hi >> 58);
    return (x >> r) | (x << ((64u - r) & 63u));
  }
  __host__ __device__ __forceinline__ double next_double() {
    return (double)(next64() >> 11) * (1.0 / 9007199254740992.0);
  }
  // pcg64_next32: low half first, high half buffered
  __host__ __device__ __forceinline__ uint32_t next32() {
    if (has32) {
      has32 = 0;
      return buf32;
    }
    uint64_t v = next64();
    has32 = 1;
    buf32 = (uint32_t)(v >> 32);
    return (uint32_t)v;
  }
  // Generator.integers(n) for n <= 2^32 (rng = n - 1 < 0xFFFFFFFF): numpy distributions.c
  // bounded_lemire_uint32 on the buffered 32-bit stream
  __host__ __device__ __forceinline__ uint32_t bounded_lemire32(uint32_t rng) {
    if (rng == 0) return 0;
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)next32() * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
      const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
      while (leftover < threshold) {
        m = (uint64_t)next32() * rng_excl;
        leftover = (uint32_t)m;
      }
    }
    return (uint32_t)(m >> 32);
  }
  // random_interval(max): uniform integer in [0, max], numpy distributions.c
  __host__ __device__ __forceinline__ uint64_t interval(uint64_t mx) {
    if (mx == 0) return 0;
    uint64_t mask = mx;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    uint64_t v;
    if (mx <= 0xffffffffull) {
      do {
        v = next32() & mask;
      } while (v > mx);
    } else {
      do {
        v = next64() & mask;
      } while (v > mx);
    }
    return v;
  }
  // pcg_setseq_128_srandom_r
  __host__ __device__ __forceinline__ void seed(U128 initstate, U128 initseq) {
    state.hi = 0;
    state.lo = 0;
    inc.hi = (initseq.hi << 1) | (initseq.lo >> 63);
    inc.lo = (initseq.lo << 1) | 1ull;
    step();
    state = add128(state, initstate);
    step();
    has32 = 0;
    buf32 = 0;
  }
  __host__ __device__ __forceinline__ void load(const uint64_t* p) {
    state.hi = p[0];
    state.lo = p[1];
    inc.hi = p[2];
    inc.lo = p[3];
    has32 = 0;
    buf32 = 0;
  }
  __host__ __device__ __forceinline__ void store(uint64_t* p) const {
    p[0] = state.hi;
    p[1] = state.lo;
    p[2] = inc.hi;
    p[3] = inc.lo;
  }
};

// ---- SeedSequence(entropy).spawn(n)[child] -> PCG64 --------------------
struct SsHash {
  uint32_t c;
  __host__ __device__ __forceinline__ uint32_t mixin(uint32_t v) {
    v ^= c;
    c *= DH_SS_MULT_A;
    v *= c;
    v ^= v >> 16;
    return v;
  }
};

__host__ __device__ __forceinline__ uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = DH_SS_MIX_L * x - DH_SS_MIX_R * y;
  r ^= r >> 16;
  return r;
}

// entropy: the parent's entropy coerced to little-endian uint32 words (host
// side does the coercion); child: index appended as the spawn_key word.
__host__ __device__ inline void seed_from_child(Pcg64& g, const uint32_t* entropy,
                                                int nwords, uint32_t child) {
  // assembled entropy = entropy (zero-padded to the pool size) ++ [child]
  const int npad = nwords < 4 ? 4 : nwords;
  const int ntot = npad + 1;
  auto word = [&](int i) -> uint32_t {
    if (i < nwords) return entropy[i];
    if (i < npad) return 0u;
    return child;
  };
  uint32_t pool[4];
  SsHash h{DH_SS_INIT_A};
  for (int i = 0; i < 4; ++i) pool[i] = h.mixin(word(i));
  for (int s = 0; s < 4; ++s)
    for (int d = 0; d < 4; ++d)
      if (s != d) pool[d] = ss_mix(pool[d], h.mixin(pool[s]));
  for (int s = 4; s < ntot; ++s)
    for (int d = 0; d < 4; ++d) pool[d] = ss_mix(pool[d], h.mixin(word(s)));
  // generate_state(4, uint64) = 8 uint32 words cycling over the pool
  uint32_t hc = DH_SS_INIT_B;
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) {
    uint32_t v = pool[i & 3];
    v ^= hc;
    hc *= DH_SS_MULT_B;
    v *= hc;
    v ^= v >> 16;
    w[i] = v;
  }
  uint64_t q[4];
  for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  U128 initstate = {q[0], q[1]};
  U128 initseq = {q[2], q[3]};
  g.seed(initstate, initseq);
}

#if defined(__HIPCC__)
// ---- ziggurat tables: staged into LDS by every kernel that draws normals ---
struct ZigLds {
  uint64_t ki[256];
  double wi[256];
  double fi[256];
};

__device__ __forceinline__ void zig_stage(ZigLds* z, const uint64_t* gki,
                                          const uint64_t* gwi, const uint64_t* gfi) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    z->ki[i] = gki[i];
    z->wi[i] = __longlong_as_double((long long)gwi[i]);
    z->fi[i] = __longlong_as_double((long long)gfi[i]);
  }
  __syncthreads();
}

#define DH_ZIG_R 3.6541528853610087963519472518
#define DH_ZIG_INV_R 0.27366123732975827203338247596

// random_standard_normal(), numpy distributions.c
__device__ __forceinline__ double std_normal(Pcg64& g, const ZigLds* z) {
// NumPy's wheel is built without FMA contraction; keep the accept/reject
// arithmetic un-fused so the decisions are the same.
#pragma clang fp contract(off)
  for (;;) {
    uint64_t r = g.next64();
    int idx = (int)(r & 0xff);
    r >>= 8;
    int sign = (int)(r & 1);
    uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
    double x = (double)rabs * z->wi[idx];
    if (sign) x = -x;
    if (rabs < z->ki[idx]) return x;  // 99.3 %
    if (idx == 0) {
      for (;;) {
        double xx = -DH_ZIG_INV_R * log1p(-g.next_double());
        double yy = -log1p(-g.next_double());
        if (yy + yy > xx * xx)
          return ((rabs >> 8) & 1) ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
      }
    } else {
      if ((z->fi[idx - 1] - z->fi[idx]) * g.next_double() + z->fi[idx] <
          exp(-0.5 * x * x))
        return x;
    }
  }
}
// ---- one generator, 64 lanes ------------------------------------------------
// A wavefront that serves ONE walker holds the walker's generator replicated in every lane.
// Drawing n values one after the other leaves 63 lanes idle behind a 128-bit multiply and an LDS
// lookup per value.  The LCG behind PCG64 jumps: the state after j steps is A_j * s + G_j * inc
// (A_j = mult^j, G_j = 1 + mult + ... + mult^(j-1)), so lane l evaluates the (l+1)-th NEXT output
// directly and the wave classifies 64 ziggurat candidates at once.  The stream is consumed exactly
// as the sequential algorithm would: candidates up to the first one that misses the fast accept
// are taken; that one is finished by the sequential wedge / tail code (wave-uniform) and the next
// round starts from the state it left.
struct PcgLanes {
  U128 A, C;  // lane l: state after l + 1 steps = A * state + C
};

__device__ __forceinline__ PcgLanes pcg_lanes_init(const Pcg64& g, int lane) {
  const U128 m = {DH_PCG_MULT_HI, DH_PCG_MULT_LO}, one = {0ull, 1ull};
  U128 A = one, G = {0ull, 0ull}, Al = one, Gl = G;
  for (int j = 1; j <= 64; ++j) {
    A = mul128(A, m);
    G = add128(mul128(G, m), one);
    if (j == lane + 1) {
      Al = A;
      Gl = G;
    }
  }
  PcgLanes L;
  L.A = Al;
  L.C = mul128(Gl, g.inc);
  return L;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t pcg_output(const U128& st) {
  const uint64_t x = st.hi ^ st.lo;
  const unsigned r = (unsigned)(st.hi >> 58);
  return (x >> r) | (x << ((64u - r) & 63u));
}

// n doubles of Generator.random() -> dst[0..n) (LDS); g advances by n.
__device__ __forceinline__ void wave_doubles(Pcg64& g, const PcgLanes& L, double* dst, int n, int lane) {
  for (int i = 0; i < n; i += 64) {
    const int m = n - i < 64 ? n - i : 64;
    const U128 st = add128(mul128(g.state, L.A), L.C);
    if (lane < m) dst[i + lane] = (double)(pcg_output(st) >> 11) * (1.0 / 9007199254740992.0);
    g.state.hi = readlane64(st.hi, m - 1);
    g.state.lo = readlane64(st.lo, m - 1);
  }
}

// n draws of random_standard_normal() -> dst[0..n) (LDS); same stream consumption as n calls of
// std_normal().  The caller synchronises LDS before reading dst.
__device__ __forceinline__ void wave_normals(Pcg64& g, const PcgLanes& L, const ZigLds* z, double* dst, int n,
                                             int lane) {
#pragma clang fp contract(off)
  int i = 0;
  while (i < n) {
    const int m = n - i < 64 ? n - i : 64;
    const U128 st = add128(mul128(g.state, L.A), L.C);
    uint64_t r = pcg_output(st);
    const int idx = (int)(r & 0xff);
    r >>= 8;
    const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
    double x = (double)rabs * z->wi[idx];
    if (r & 1) x = -x;
    const bool fast = rabs < z->ki[idx];
    const uint64_t miss = __ballot(!fast && lane < m);
    const int f = miss ? __ffsll((long long)miss) - 1 : m;  // candidates 0..f-1 are accepted as they are
    if (lane < f) dst[i + lane] = x;
    i += f;
    const int last = miss ? f : m - 1;
    g.state.hi = readlane64(st.hi, last);
    g.state.lo = readlane64(st.lo, last);
    if (miss) {
      // candidate f: wedge or tail, with the uniforms that follow it in the stream
      const int idf = __builtin_amdgcn_readlane(idx, f);
      const uint64_t rabsf = readlane64(rabs, f);
      double xf = __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(x), f));
      bool take;
      if (idf == 0) {
        for (;;) {
          const double xx = -DH_ZIG_INV_R * log1p(-g.next_double());
          const double yy = -log1p(-g.next_double());
          if (yy + yy > xx * xx) {
            xf = ((rabsf >> 8) & 1) ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
            break;
          }
        }
        take = true;
      } else {
        take = (z->fi[idf - 1] - z->fi[idf]) * g.next_double() + z->fi[idf] < exp(-0.5 * xf * xf);
      }
      if (take) {
        if (lane == 0) dst[i] = xf;
        ++i;
      }
    }
  }
}

// nc standard normals -> column dst[i * 64 + lane] (LDS), returns sum of squares.
// Same stream consumption as nc calls of std_normal(), but software-pipelined:
// the PCG step and the ziggurat table lookups of the NEXT candidate are issued
// before the current one is classified, so the LDS latency of the lookup and the
// 128-bit multiply overlap; a rejected candidate consumes the prefetched draw
// as its wedge/tail uniform exactly as the sequential algorithm would, and a
// lane that is finished hands its unconsumed candidate back (state rewind).
#ifdef DH_NORMALS_NOINLINE
#define DH_NORMALS_INLINE __attribute__((noinline))
#else
#define DH_NORMALS_INLINE __forceinline__
#endif
__device__ DH_NORMALS_INLINE double normals_to_lds(Pcg64& g, const ZigLds* z, double* dst, int lane,
                                                   int nc) {
#pragma clang fp contract(off)
  double ss = 0.0;
  int i = 0;
  U128 sb = g.state;  // generator state before `cand` was drawn
  uint64_t cand = g.next64();
  uint64_t cki = z->ki[cand & 0xff];
  double cwi = z->wi[cand & 0xff];
  while (__any(i < nc)) {
    const bool active = i < nc;
    const U128 sn = g.state;  // state before `nxt`
    const uint64_t nxt = g.next64();
    const uint64_t nki = z->ki[nxt & 0xff];
    const double nwi = z->wi[nxt & 0xff];
    if (active) {
      const int idx = (int)(cand & 0xff);
      const uint64_t rabs = (cand >> 9) & 0x000fffffffffffffull;
      // (double)rabs, exact for rabs < 2^52, as 2^52 + rabs with the integer in the mantissa field (two
      // instructions instead of cvt / ldexp / cvt / add); the sign is bit 8 of the draw, moved onto x's
      // sign bit (x >= 0 here; +0 becomes -0 exactly as `x = -x` would make it)
      const double rd = __longlong_as_double((long long)(rabs | 0x4330000000000000ull)) - 4503599627370496.0;
      double x = rd * cwi;
      x = __longlong_as_double(__double_as_longlong(x) ^ (long long)((cand & 0x100ull) << 55));
      bool take = rabs < cki;
      bool used_next = false;
      if (!take) {
        used_next = true;
        double u1 = (double)(nxt >> 11) * (1.0 / 9007199254740992.0);
        if (idx == 0) {
          for (;;) {
            const double xx = -DH_ZIG_INV_R * log1p(-u1);
            const double yy = -log1p(-g.next_double());
            if (yy + yy > xx * xx) {
              x = ((rabs >> 8) & 1) ? -(DH_ZIG_R + xx) : DH_ZIG_R + xx;
              break;
            }
            u1 = g.next_double();
          }
          take = true;
        } else {
          take = (z->fi[idx - 1] - z->fi[idx]) * u1 + z->fi[idx] < exp(-0.5 * x * x);
        }
      }
      if (take) {
        dst[i * 64 + lane] = x;
        ss = fma(x, x, ss);
        ++i;
      }
      if (used_next) {
        sb = g.state;
        cand = g.next64();
        cki = z->ki[cand & 0xff];
        cwi = z->wi[cand & 0xff];
      } else {
        sb = sn;
        cand = nxt;
        cki = nki;
        cwi = nwi;
      }
      if (i >= nc) g.state = sb;  // done: give the unconsumed candidate back
    } else {
      g.state = sn;  // idle lane: undo the speculative step
    }
  }
  return ss;
}
#endif  // __HIPCC__

}  // namespace dh
