// Generator policies of the proposal kernels.
//
//   RNG_PCG64   numpy.random.Generator(PCG64) streams, bit for bit (rng_pcg64.h): the parity mode -- same seed,
//               same proposals as the reference (utils.py:993-1009).
//   RNG_PHILOX  hiprand's Philox4x32-10 device generator (hiprand_kernel.h), keyed (seed, subsequence = seq0 +
//               walker, offset): counter based, no generator state in HBM.  The throughput mode (north_star:
//               "hiprand for the unit-cube draws"): same algorithms, same distributions -- normals are hiprand's
//               fp32 Box-Muller values widened to fp64, uniforms 53-bit -- but not the reference's streams, so it
//               is validated statistically (tests/test_gpu_philox.py).
//
// LaneGen<RNG>: one stream per lane (walker-per-lane kernels: walk.hip, walk2.hip).
// WaveGen<RNG>: one stream per wavefront, vector draws produced by all lanes (wave-per-walker kernels: wide.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>
#include <stdint.h>

#include "ctx.h"  // PhiloxKey
#include "rng_pcg64.h"

namespace dh {

enum : int { RNG_PCG64 = 0, RNG_PHILOX = 1 };

#if defined(__HIPCC__)
template <int RNG>
struct LaneGen;

template <>
struct LaneGen<RNG_PCG64> {
  Pcg64 g;
  const ZigLds* z;
  __device__ __forceinline__ void init(const uint64_t* rng_in, size_t wi, const ZigLds* zz, const PhiloxKey&) {
    g.load(rng_in + wi * 4);
    z = zz;
  }
  __device__ __forceinline__ double uniform() { return g.next_double(); }
  __device__ __forceinline__ double normal() { return std_normal(g, z); }
  __device__ __forceinline__ uint64_t interval(uint64_t mx) { return g.interval(mx); }
  __device__ __forceinline__ uint32_t bounded32(uint32_t rng) { return g.bounded_lemire32(rng); }
  __device__ __forceinline__ void store(uint64_t* out, size_t w) const {
    if (out) g.store(out + w * 4);
  }
};

template <>
struct LaneGen<RNG_PHILOX> {
  hiprandStatePhilox4_32_10_t st;
  __device__ __forceinline__ void init(const uint64_t*, size_t wi, const ZigLds*, const PhiloxKey& k) {
    hiprand_init(k.seed, k.seq0 + (unsigned long long)wi, k.offset, &st);
  }
  // [0, 1) like Generator.random() (hiprand's doubles live in (0, 1])
  __device__ __forceinline__ double uniform() { return 1.0 - hiprand_uniform_double(&st); }
  __device__ __forceinline__ double normal() { return (double)hiprand_normal(&st); }
  // uniform integer in [0, mx], mx < 2^32: masked rejection (the scheme of numpy's random_interval)
  __device__ __forceinline__ uint64_t interval(uint64_t mx) {
    if (mx == 0) return 0;
    uint32_t mask = (uint32_t)mx;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
      v = hiprand(&st) & mask;
    } while (v > (uint32_t)mx);
    return v;
  }
  __device__ __forceinline__ uint32_t bounded32(uint32_t rng) { return (uint32_t)interval(rng); }
  __device__ __forceinline__ void store(uint64_t*, size_t) const {}
};

// ---- one stream, 64 lanes ----------------------------------------------------------------------------
template <int RNG>
struct WaveGen;

template <>
struct WaveGen<RNG_PCG64> {
  Pcg64 g;
  PcgLanes PL;
  const ZigLds* z;
  __device__ __forceinline__ void init(const uint64_t* rng_in, size_t w, int lane, const ZigLds* zz, const PhiloxKey&) {
    g.load(rng_in + w * 4);
    PL = pcg_lanes_init(g, lane);
    z = zz;
  }
  __device__ __forceinline__ double uniform() { return g.next_double(); }
  __device__ __forceinline__ uint64_t interval(uint64_t mx) { return g.interval(mx); }
  __device__ __forceinline__ void doubles(double* dst, int n, int lane) { wave_doubles(g, PL, dst, n, lane); }
  __device__ __forceinline__ void normals(double* dst, int n, int lane) { wave_normals(g, PL, z, dst, n, lane); }
  __device__ __forceinline__ void store(uint64_t* out, size_t w) const {
    if (out) g.store(out + w * 4);
  }
};

// Counter based: a draw at stream position p is a function of (seed, subsequence, p), so the lanes of the wave
// produce a vector's entries independently -- lane l the block at pos + 4 l -- and scalar draws are computed
// redundantly by every lane (wave-uniform), one block each.  `pos` advances in whole blocks of four 32-bit draws.
template <>
struct WaveGen<RNG_PHILOX> {
  unsigned long long seed, seq, pos;
  __device__ __forceinline__ void init(const uint64_t*, size_t w, int, const ZigLds*, const PhiloxKey& k) {
    seed = k.seed;
    seq = k.seq0 + (unsigned long long)w;
    pos = k.offset;
  }
  __device__ __forceinline__ double uniform() {
    hiprandStatePhilox4_32_10_t st;
    hiprand_init(seed, seq, pos, &st);
    pos += 4;
    return 1.0 - hiprand_uniform_double(&st);
  }
  __device__ __forceinline__ uint64_t interval(uint64_t mx) {
    if (mx == 0) return 0;
    uint32_t mask = (uint32_t)mx;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    for (;;) {
      hiprandStatePhilox4_32_10_t st;
      hiprand_init(seed, seq, pos, &st);
      pos += 4;
      for (int i = 0; i < 4; ++i) {  // the four draws of the block, in order
        const uint32_t v = hiprand(&st) & mask;
        if (v <= (uint32_t)mx) return v;
      }
    }
  }
  // n doubles in [0, 1) -> dst[0..n) (LDS): lane l the pair 2 l, 2 l + 1 of every 128
  __device__ __forceinline__ void doubles(double* dst, int n, int lane) {
    for (int i0 = 0; i0 < n; i0 += 128) {
      const int i = i0 + 2 * lane;
      if (i < n) {
        hiprandStatePhilox4_32_10_t st;
        hiprand_init(seed, seq, pos + 2ull * i, &st);
        const double2 d = hiprand_uniform2_double(&st);
        dst[i] = 1.0 - d.x;
        if (i + 1 < n) dst[i + 1] = 1.0 - d.y;
      }
    }
    pos += 4ull * ((n + 1) / 2);
  }
  // n standard normals -> dst[0..n) (LDS): lane l the four 4 l .. 4 l + 3 of every 256
  __device__ __forceinline__ void normals(double* dst, int n, int lane) {
    for (int i0 = 0; i0 < n; i0 += 256) {
      const int i = i0 + 4 * lane;
      if (i < n) {
        hiprandStatePhilox4_32_10_t st;
        hiprand_init(seed, seq, pos + (unsigned long long)i, &st);
        const float4 zf = hiprand_normal4(&st);
        dst[i] = (double)zf.x;
        if (i + 1 < n) dst[i + 1] = (double)zf.y;
        if (i + 2 < n) dst[i + 2] = (double)zf.z;
        if (i + 3 < n) dst[i + 3] = (double)zf.w;
      }
    }
    pos += 4ull * ((n + 3) / 4);
  }
  __device__ __forceinline__ void store(uint64_t*, size_t) const {}
};
#endif  // __HIPCC__

}  // namespace dh
