// Shared pieces of the fused edge-layer kernel (edge_layer.hip): 16-bit element traits,
// fp32 -> two-plane split, LDS swizzle of the weight stages, fast sigmoid.
#pragma once
#include "common.h"
#include "kernels.h"

namespace difusco {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// kScaled: the 16-bit format has a narrow exponent range (fp16: normal numbers 2^-14 .. 65504), so operands are brought
// into its upper binades by exact power-of-two scales before they are split (weights: on the host, weights.py;
// activations: in registers) and the scale is undone where the bias is added.  bf16 has the fp32 exponent range.
struct FBf16 {
  static constexpr bool kScaled = false;
  typedef v8bf frag;
  __device__ static __forceinline__ unsigned split_pair(float& a, float& b) {
    v2f f = {a, b};
    v2bf h = __builtin_convertvector(f, v2bf);
    f = f - __builtin_convertvector(h, v2f);      // (one packed subtract for the pair)
    a = f[0];
    b = f[1];
    return __builtin_bit_cast(unsigned, h);
  }
  __device__ static __forceinline__ v16f mfma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
struct FFp16 {
  static constexpr bool kScaled = true;
  typedef v8h frag;
  __device__ static __forceinline__ unsigned split_pair(float& a, float& b) {
    v2f f = {a, b};
    v2h h = __builtin_convertvector(f, v2h);
    f = f - __builtin_convertvector(h, v2f);      // (one packed subtract for the pair)
    a = f[0];
    b = f[1];
    return __builtin_bit_cast(unsigned, h);
  }
  __device__ static __forceinline__ v16f mfma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// eight fp32 -> two planes of eight 16-bit values (hi, lo)
template <typename T>
__device__ __forceinline__ void split8(const float (&x)[8], typename T::frag& hi, typename T::frag& lo) {
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = x[q];
  v4u h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) h[q] = T::split_pair(v[2 * q], v[2 * q + 1]);
#pragma unroll
  for (int q = 0; q < 4; ++q) l[q] = T::split_pair(v[2 * q], v[2 * q + 1]);
  hi = __builtin_bit_cast(typename T::frag, h);
  lo = __builtin_bit_cast(typename T::frag, l);
}

// LDS element offset of (entry, half) inside a plane: 32-byte rows, halves swapped on odd 8-row groups
__device__ __forceinline__ int wslot(int entry, int half) { return entry * 16 + ((half ^ ((entry >> 3) & 1)) << 3); }

#define DIFUSCO_PAIR(v, i) (v2f{(v)[(i)], (v)[(i) + 1]})
// sigmoid(x) from xl = x log2(e): 1 / (1 + exp2(-xl)).  The negation is a source modifier of v_exp_f32: add + two transcendentals.
__device__ __forceinline__ v2f sigmoid2_log2e(v2f xl) {
  v2f t = v2f{__builtin_amdgcn_exp2f(-xl[0]), __builtin_amdgcn_exp2f(-xl[1])};
  t = v2f{1.0f, 1.0f} + t;
  return v2f{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
}
// SiLU(z) log2(e) / s from zl = z log2(e):  zl / ((1 + exp2(-zl)) s)  - the operand scale 1 / s = 2^ka of GEMM 2 rides in the
// multiply-add that forms the denominator (s is a power of two: t s + s is the exact (1 + t) s, rounded once).
// exp2 -> inf (zl < -128): the denominator is inf, its reciprocal 0, the result -0 - like SiLU itself rounds.
__device__ __forceinline__ v2f silu2_log2e(v2f zl, float s) {
  v2f t = v2f{__builtin_amdgcn_exp2f(-zl[0]), __builtin_amdgcn_exp2f(-zl[1])};
  t = t * v2f{s, s} + v2f{s, s};
  return zl * v2f{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
}

}  // namespace difusco
