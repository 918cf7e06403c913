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
  // (no bf16 form of the mixed-precision multiply-add: the plain sequence)
  __device__ static __forceinline__ unsigned mul_hi_pair(float a, float sa, float b, float sb) {
    float pa = a * sa, pb = b * sb;
    return split_pair(pa, pb);
  }
  __device__ static __forceinline__ unsigned mul_lo_pair(float a, float sa, float b, float sb, unsigned) {
    float pa = a * sa, pb = b * sb;
    split_pair(pa, pb);
    return split_pair(pa, pb);
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
  // mul_hi_pair / mul_lo_pair: the two planes of the PRODUCTS a sa, b sb in four instructions (VOP3P-MIX: v_fma_mix{lo,hi}_f16 computes an fp32
  // multiply-add of fp32 / fp16 sources and writes the result, rounded to fp16, into one half of the destination):
  //   hi = fp16(a sa), fp16(b sb);   lo = fp16(a sa - hi), fp16(b sb - hi)      (hi read back as the fp16 source of the add)
  // instead of the six of "multiply, split_pair" (2 v_mul, v_cvt_pk, 2 v_fma_mix_f32, v_cvt_pk).  With a power-of-two
  // multiplier (the operand scale of GEMM 1) the product is exact and both planes have the bits of the six-instruction
  // form; with a general multiplier (SiLU: z sigmoid(z)) the product is never rounded to fp32 on its own - hi + lo then
  // carries it to the same 22 bits, a different last bit of lo in a fraction of the elements.
  // (two calls, so that split8_mul can issue the hi halves of all its pairs before the first lo half: an instruction that
  //  reads a register right after a write to one of its 16-bit halves costs a wait state on gfx950)
  __device__ static __forceinline__ unsigned mul_hi_pair(float a, float sa, float b, float sb) {
    unsigned h = 0;
#if defined(__HIP_DEVICE_COMPILE__)      // (the "v" constraint does not exist on the host pass)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(sa));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(sb));
#endif
    return h;
  }
  __device__ static __forceinline__ unsigned mul_lo_pair(float a, float sa, float b, float sb, unsigned h) {
    unsigned l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(sa), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(sb), "v"(h));
#endif
    return l;
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

// the planes of the eight products x[q] m[q] (T::mul_hi_pair / mul_lo_pair)
template <typename T>
__device__ __forceinline__ void split8_mul(const float (&x)[8], const float (&m)[8], typename T::frag& hi, typename T::frag& lo) {
  v4u h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) h[q] = T::mul_hi_pair(x[2 * q], m[2 * q], x[2 * q + 1], m[2 * q + 1]);
#pragma unroll
  for (int q = 0; q < 4; ++q) l[q] = T::mul_lo_pair(x[2 * q], m[2 * q], x[2 * q + 1], m[2 * q + 1], h[q]);
  hi = __builtin_bit_cast(typename T::frag, h);
  lo = __builtin_bit_cast(typename T::frag, l);
}

// LDS element offset of (entry, half) inside a plane: 32-byte rows, halves swapped on odd 8-row groups
__device__ __forceinline__ int wslot(int entry, int half) { return entry * 16 + ((half ^ ((entry >> 3) & 1)) << 3); }

// the same on a register pair: the multiply and the add are packed (v_pk_mul_f32, v_pk_add_f32); element for element the
// arithmetic of fast_sigmoid
__device__ __forceinline__ v2f fast_sigmoid2(v2f x) {
  v2f t = x * v2f{-1.4426950408889634f, -1.4426950408889634f};
  t = v2f{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  t = v2f{1.0f, 1.0f} + t;
  return v2f{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
}
#define DIFUSCO_PAIR(v, i) (v2f{(v)[(i)], (v)[(i) + 1]})
// sigmoid(x) for an argument that arrives pre-scaled, xs = x * 2^k: nsig = -log2(e) * 2^-k (a power-of-two multiple of the
// constant above, so xs * nsig has the bits of x * -log2(e))
__device__ __forceinline__ v2f fast_sigmoid2s(v2f xs, float nsig) {
  v2f t = xs * v2f{nsig, nsig};
  t = v2f{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  t = v2f{1.0f, 1.0f} + t;
  return v2f{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
}
__device__ __forceinline__ float fast_sigmoid_s(float xs, float nsig) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(nsig * xs));
}

}  // namespace difusco
