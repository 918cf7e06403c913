// Device-side helpers shared by the gfx950 kernels.  CDNA4 only: wavefront = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace difusco {

constexpr int kWave = 64;

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Butterfly all-reduce over the 64 lanes of a wavefront: every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Two independent sums reduced together (the two chains interleave -> latency of one).
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float ta = __shfl_xor(a, off, kWave);
    float tb = __shfl_xor(b, off, kWave);
    a += ta;
    b += tb;
  }
}

// Power-of-two scale that brings a non-negative maximum m into [2^14, 2^15): returns 2^k and writes 2^-k, k clamped to
// [-100, 100] (m = 0, denormal or huge: the clamp; inf / nan inputs give garbage downstream either way).  Integer
// arithmetic on the exponent field only.
__device__ __forceinline__ float pow2_scale_for(float m, float& inv) {
  int ex = (int)(__builtin_bit_cast(unsigned, m) >> 23) & 255;     // biased exponent of m
  ex = ex < 41 ? 41 : (ex > 241 ? 241 : ex);
  const int k = 141 - ex;                                          // m * 2^k in [2^14, 2^15)
  inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
  return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

// sigmoid on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each)
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: one call gives 4 x 32 random bits -------
struct Philox {
  static constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u;
  static constexpr uint32_t kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
  __device__ static inline void run(uint64_t seed, uint64_t offset, uint64_t index, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)index, c1 = (uint32_t)(index >> 32);
    uint32_t c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0 = __umulhi(kM0, c0), lo0 = kM0 * c0;
      uint32_t hi1 = __umulhi(kM1, c2), lo1 = kM1 * c2;
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += kW0; k1 += kW1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
  // uniform in [0,1) with 24 random bits (exactly representable in fp32)
  __device__ static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
};

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t index) {
  uint32_t r[4];
  Philox::run(seed, offset, index, r);
  return Philox::u01(r[0]);
}

__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t index) {
  uint32_t r[4];
  Philox::run(seed, offset, index, r);
  // Box-Muller; u1 in (0,1] so the log is finite
  float u1 = ((float)(r[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
  float u2 = Philox::u01(r[1]);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace difusco
