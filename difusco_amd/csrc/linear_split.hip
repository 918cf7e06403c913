// Split-precision row-linear on the gfx950 bf16 matrix cores:  Y = X * W^T + b (+ residual), fp32 in and
// out, each fp32 operand decomposed into NS bf16 planes (x = hi + mid (+ lo), round-to-nearest-even):
//   NS = 2 -> 3 MFMA products  (hi*hi, hi*mid, mid*hi)                       ~2^-17 relative per product
//   NS = 3 -> 6 MFMA products  (+ hi*lo, lo*hi, mid*mid): 8+8+8 = 24 mantissa bits, i.e. the full fp32
//             significand of both operands; dropped terms are <= 2^-24 relative -> fp32-class accuracy.
// Products of bf16 pairs are exact in the fp32 accumulator, so the only rounding is the fp32 accumulation
// itself.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md),
// so 6 products cost 6/16 of the exact-fp32 MFMA path and the E-row linears of the DIFUSCO layer
// (difusco/models/gnn_encoder.py:104 `C`, :344 `per_layer_out[2]`, :395 `edge_embed`) become HBM-bound.
//
// A third mode decomposes into TWO fp16 planes (11+11 significand bits, 3 products): fp32-class accuracy
// at half the matrix-core work of the 6-product bf16 mode.  fp16 has 5 exponent bits, so both operands are
// first scaled by exact powers of two into [2^14, 2^15) at their maximum (weights per matrix / per row on the
// host, rows of x by SplitScale::row_scale) and the product of the inverse scales is applied where
// the bias is added - any finite fp32 operand keeps its 22 bits.
// The node-row shape of a layer ([N,256] x [256,1024]) has its own kernel (node_linear.hip); this one serves the
// E-row linears of the unfused path and the remaining shapes.
//
// Inside every 16-wide k slab the two middle groups of 4 are swapped (slab position j holds
// k = {0..3, 8..11, 4..7, 12..15}[j]) - the order in which a 32x32 MFMA accumulator hands 8 of its
// registers to the next MFMA as a B operand (used by the fused layer kernel, edge_layer.hip).
//
// Same orientation as linear.hip (transposed: A = 32 weight rows, B = 32 data rows; a lane owns 4
// consecutive output features of one data row per accumulator quad).  Weights arrive PRE-SPLIT from the
// host (weights.py) as planes laid out [K/16 slabs][n_out rows][16 k] bf16, so a slab is one linear
// stream; X is split on the fly while it is staged into LDS.
#include "common.h"
#include "kernels.h"

namespace difusco {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

// 16-bit element traits: (a, b) -> packed pair (RNE) + fp32 remainders; the matching MFMA
struct Bf16 {
  typedef v8bf frag;
  __device__ static __forceinline__ unsigned split_pair(float& a, float& b) {
    v2f f = {a, b};
    v2bf h = __builtin_convertvector(f, v2bf);
    a -= (float)h[0];
    b -= (float)h[1];
    return __builtin_bit_cast(unsigned, h);
  }
  __device__ static __forceinline__ v16f mfma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
struct Fp16 {
  typedef v8h frag;
  __device__ static __forceinline__ unsigned split_pair(float& a, float& b) {
    v2f f = {a, b};
    v2h h = __builtin_convertvector(f, v2h);
    a -= (float)h[0];
    b -= (float)h[1];
    return __builtin_bit_cast(unsigned, h);
  }
  __device__ static __forceinline__ v16f mfma(frag a, frag b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// GEN: the rows of X are not read but generated - ScalarEmbeddingSine of one scalar per row (gnn_encoder.py:230-249):
// X[r][c] = sin / cos (x[r] / dim_t[c]) for even / odd c, the arithmetic of scalar_embed_kernel.  X then carries the
// scalars (gen_x, indexed through gen_perm) and the E x H embedding never exists in memory.
// D: how many k steps ahead the global loads of a step are issued (register ring of D slots, loop fully unrolled).
// D = 1 for the E-row linears (thousands of workgroups hide the latency).  The node-row linear has one or two workgroups
// per CU and its weight planes come from MALL (the e stream of the edge kernels sweeps the L2 between two layers), so with
// D = 1 every one of its 16 k steps waits for a ~2,000-cycle round trip: D = 4 keeps four steps in flight.
template <int K, int FB, int NS, typename T, bool GEN = false, int D = 1>
__global__ __launch_bounds__(256, 2) void linear_rows_split_kernel(const float* __restrict__ X,
                                                                    const unsigned short* __restrict__ Wp,
                                                                    long long plane_stride,  // elements between planes
                                                                    int n_out_total,
                                                                    const float* __restrict__ bias,
                                                                    const float* residual, float* Y, long long M,
                                                                    long long ldy, int tiled_out,
                                                                    const int* __restrict__ gen_perm,
                                                                    const float* __restrict__ gen_dimt,
                                                                    const float* __restrict__ row_scale, float x_scale,
                                                                    const float* __restrict__ w_inv,
                                                                    float* __restrict__ tile_max) {
  // Operand scaling for the fp16 planes (exact powers of two, see edge_layer_common.h): row r of X is multiplied by
  // x_scale * row_scale[r] before it is split (row_scale: what the producer of X left per row, or null), the weight
  // planes hold W[f] / w_inv[f] (weights.py), and the accumulator is multiplied by the product of the inverses where the
  // bias is added.  tile_max (tiled output only): max |Y| per 32-row tile, the e-stream scale of the fused edge kernel.
  constexpr int RB = 128, NB = FB / 32, BK = 16;
  constexpr int RS = 24;                 // LDS row stride in bf16 elements (32 B data + 16 B pad = 48 B)
  constexpr int WV = (FB * 2 + 255) / 256;  // 16-byte chunks per thread per weight plane per slab
  static_assert(K % BK == 0, "K must be a multiple of 16");
  static_assert(NS == 2 || NS == 3, "2 or 3 planes");

  extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
  unsigned short* Xs = smem_s;                         // [NS][RB][RS]
  unsigned short* Ws = smem_s + NS * RB * RS;          // [NS][FB][RS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  // D > 1 (node-row linear): 1-D grid mapped so that the column blocks of one row block run on ONE XCD (workgroup b runs on
  // XCD b % 8), back to back - the X rows are then fetched into one L2 once instead of into eight.
  int rb_i = blockIdx.x, cb_i = blockIdx.y;
  if constexpr (D > 1) {
    const int ncb = n_out_total / FB, slot = (int)blockIdx.x >> 3;
    rb_i = (slot / ncb) * 8 + ((int)blockIdx.x & 7);
    cb_i = slot % ncb;
    if ((long long)rb_i * RB >= M) return;
  }
  const long long r0 = (long long)rb_i * RB;
  const int f0 = cb_i * FB;

  v16f acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  v4f xr_[D][2];
  v4u wr_[D][NS][WV];
  float xsc[2];      // operand scale of the two X rows this thread stages
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long gr = r0 + ((tid + 256 * i) >> 2);
    gr = gr < M ? gr : M - 1;
    xsc[i] = x_scale * (row_scale != nullptr ? row_scale[gr] : 1.0f);
  }

#define DIFUSCO_LOAD(KT, SLOT)                                                                            \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      const int idx = tid + 256 * i;                                                                  \
      const int row = idx >> 2, c4 = idx & 3;                                                         \
      long long gr = r0 + row;                                                                        \
      gr = gr < M ? gr : M - 1;                                                                       \
      if constexpr (GEN) {                                                                            \
        const float xv = X[gen_perm ? gen_perm[gr] : gr];                                             \
        /* features 2j, 2j + 1 are sin, cos of the SAME angle (dim_t[2j] == dim_t[2j + 1], gnn_encoder.py:242-247): one */ \
        /* precise sincosf - one argument reduction - per pair instead of a sinf and a cosf (round 5) */       \
        _Pragma("unroll") for (int q2 = 0; q2 < 2; ++q2) {                                            \
          const int c = (KT) + c4 * 4 + 2 * q2;                                                       \
          const float v = xv / gen_dimt[c];                                                           \
          float sn, cs;                                                                               \
          sincosf(v, &sn, &cs);                                                                       \
          xr_[SLOT][i][2 * q2] = sn;                                                                  \
          xr_[SLOT][i][2 * q2 + 1] = cs;                                                              \
        }                                                                                             \
      } else {                                                                                        \
        xr_[SLOT][i] = *reinterpret_cast<const v4f*>(X + gr * K + (KT) + c4 * 4);                            \
      }                                                                                               \
    }                                                                                                 \
    const unsigned short* wslab = Wp + ((long long)((KT) / BK) * n_out_total + f0) * BK;              \
    _Pragma("unroll") for (int p = 0; p < NS; ++p) {                                                  \
      _Pragma("unroll") for (int i = 0; i < WV; ++i) {                                                \
        int c = tid + 256 * i;                                                                        \
        if (FB * 2 % 256 != 0) c = c < FB * 2 ? c : FB * 2 - 1;                                       \
        wr_[SLOT][p][i] = *reinterpret_cast<const v4u*>(wslab + p * plane_stride + (long long)c * 8);        \
      }                                                                                               \
    }                                                                                                 \
  }
#define DIFUSCO_STORE(SLOT)                                                                           \
  {                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      const int idx = tid + 256 * i;                                                                  \
      const int row = idx >> 2, c4 = idx & 3;                                                         \
      float a = xr_[SLOT][i][0] * xsc[i], b = xr_[SLOT][i][1] * xsc[i], c = xr_[SLOT][i][2] * xsc[i],  \
            d = xr_[SLOT][i][3] * xsc[i];                                                               \
      _Pragma("unroll") for (int p = 0; p < NS; ++p) {                                                \
        v2u pk;                                                                                       \
        pk[0] = T::split_pair(a, b);                                                                  \
        pk[1] = T::split_pair(c, d);                                                                  \
        /* k group c4 goes to slab position group {0,2,1,3}[c4] */                                    \
        *reinterpret_cast<v2u*>(Xs + (p * RB + row) * RS + (((c4 & 1) << 1) | (c4 >> 1)) * 4) = pk;   \
      }                                                                                               \
    }                                                                                                 \
    _Pragma("unroll") for (int p = 0; p < NS; ++p) {                                                  \
      _Pragma("unroll") for (int i = 0; i < WV; ++i) {                                                \
        int c = tid + 256 * i;                                                                        \
        if (FB * 2 % 256 != 0) c = c < FB * 2 ? c : FB * 2 - 1;                                       \
        *reinterpret_cast<v4u*>(Ws + (p * FB + (c >> 1)) * RS + (c & 1) * 8) = wr_[SLOT][p][i];             \
      }                                                                                               \
    }                                                                                                 \
  }

  constexpr int NSTEP = K / BK;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < NSTEP) DIFUSCO_LOAD(d * BK, d)
  DIFUSCO_STORE(0)
  __syncthreads();

  const unsigned short* xrow = Xs + (wave * 32 + l31) * RS + hh * 8;
  const unsigned short* wrow = Ws + l31 * RS + hh * 8;

#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    if (st + D < NSTEP) DIFUSCO_LOAD((st + D) * BK, st % D)      // slot st % D was parked in LDS before this step
    typedef typename T::frag frag;
    frag xb[NS];
#pragma unroll
    for (int p = 0; p < NS; ++p) xb[p] = *reinterpret_cast<const frag*>(xrow + p * RB * RS);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      frag wa[NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) wa[p] = *reinterpret_cast<const frag*>(wrow + (p * FB + nb * 32) * RS);
      // smallest terms first
      if constexpr (NS == 3) {
        acc[nb] = T::mfma(wa[2], xb[0], acc[nb]);
        acc[nb] = T::mfma(wa[0], xb[2], acc[nb]);
        acc[nb] = T::mfma(wa[1], xb[1], acc[nb]);
      }
      acc[nb] = T::mfma(wa[1], xb[0], acc[nb]);
      acc[nb] = T::mfma(wa[0], xb[1], acc[nb]);
      acc[nb] = T::mfma(wa[0], xb[0], acc[nb]);
    }
    if (st + 1 < NSTEP) {
      __syncthreads();
      DIFUSCO_STORE((st + 1) % D)
      __syncthreads();
    }
  }
#undef DIFUSCO_LOAD
#undef DIFUSCO_STORE

  const long long row = r0 + wave * 32 + l31;
  float tmx = 0.0f;
  if (row < M) {
    float rinv = 1.0f;      // 1 / (x_scale * row_scale[row]): exact, both are powers of two
    {
      const float sc = x_scale * (row_scale != nullptr ? row_scale[row] : 1.0f);
      rinv = __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, sc));
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = f0 + nb * 32 + 8 * g + 4 * hh;
        v4f v = {acc[nb][4 * g + 0], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
        if (w_inv != nullptr) v = v * (*reinterpret_cast<const v4f*>(w_inv + f) * rinv);
        else v = v * rinv;
        if (bias != nullptr) v += *reinterpret_cast<const v4f*>(bias + f);
        if (residual != nullptr) v += *reinterpret_cast<const v4f*>(residual + row * ldy + f);
        if (tiled_out) *reinterpret_cast<v4f*>(Y + edge_tiled_offset(row, f)) = v;   // f % 4 == 0: one aligned float4
        else *reinterpret_cast<v4f*>(Y + row * ldy + f) = v;
        tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
        tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
      }
    }
  }
  if (tile_max != nullptr) {      // (FB == n_out: this wave holds every feature of its 32 rows)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tmx = __builtin_fmaxf(tmx, __shfl_xor(tmx, off, 64));
    if (lane == 0) tile_max[(r0 >> 5) + wave] = tmx;
  }
}

template <int K, int FB, int NS, typename T, bool GEN = false, int D = 1>
static hipError_t launch_split(const float* x, const unsigned short* wp, long long plane_stride, const float* bias,
                               const float* residual, float* y, long long m, int n_out, long long ldy, hipStream_t stream,
                               int tiled_out, const int* gen_perm, const float* gen_dimt, const float* row_scale,
                               float x_scale, const float* w_inv, float* tile_max) {
  constexpr size_t lds = (size_t)NS * (128 + FB) * 24 * sizeof(unsigned short);
  static std::atomic<unsigned long long> attr_devices{0};
  {
    hipError_t e = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&linear_rows_split_kernel<K, FB, NS, T, GEN, D>), (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((m + 127) / 128), (unsigned)(n_out / FB));
  if (D > 1) grid = dim3((unsigned)(8 * (((m + 127) / 128 + 7) / 8) * (n_out / FB)), 1);
  hipLaunchKernelGGL((linear_rows_split_kernel<K, FB, NS, T, GEN, D>), grid, dim3(256), lds, stream, x, wp, plane_stride, n_out,
                     bias, residual, y, m, ldy, tiled_out, gen_perm, gen_dimt, row_scale, x_scale, w_inv, tile_max);
  return hipGetLastError();
}

#ifdef DIFUSCO_PROFILING
// A/B knob of the profiling library (difusco_debug_set key 8): 0 = node_linear.hip (production), 4 / 1 = this file's kernel with
// 4 / 1 k steps of lookahead (rounds 2 / 1)
int g_node_linear_depth = 0;
#define NODE_LINEAR_DEPTH g_node_linear_depth
#else
#define NODE_LINEAR_DEPTH 0
#endif

// wp: first plane of the chosen element type, plane p at wp + p*plane_stride, each [K/16][n_out][16]
// (k-permuted).  mode: 1 = bf16 x 2 planes (3 products), 2 = bf16 x 3 planes (6 products),
// 3 = fp16 x 2 planes (3 products).  sc: operand scaling of the fp16 path (kernels.h: SplitScale).
hipError_t linear_rows_split(const float* x, const unsigned short* wp, long long plane_stride, int mode,
                             const float* bias, const float* residual, float* y, long long m, int k, int n_out,
                             long long ldy, hipStream_t stream, int tiled_out, SplitScale sc) {
  if (m <= 0) return hipSuccess;
  if (mode < 1 || mode > 3) return hipErrorInvalidValue;
  if (tiled_out && (k != 256 || n_out != 256 || residual != nullptr)) return hipErrorInvalidValue;
  if (sc.tile_max != nullptr && !tiled_out) return hipErrorInvalidValue;
#define DIFUSCO_SPLIT_ARGS x, wp, plane_stride, bias, residual, y, m, n_out, ldy, stream, tiled_out, nullptr, nullptr, sc.row_scale, sc.x_scale, sc.w_inv, sc.tile_max
#define DIFUSCO_SPLIT_CASE(KK, FBB)                                                   \
  if (k == KK && n_out % FBB == 0) {                                                  \
    if (mode == 1) return launch_split<KK, FBB, 2, Bf16>(DIFUSCO_SPLIT_ARGS);         \
    if (mode == 2) return launch_split<KK, FBB, 3, Bf16>(DIFUSCO_SPLIT_ARGS);         \
    return launch_split<KK, FBB, 2, Fp16>(DIFUSCO_SPLIT_ARGS);                        \
  }
  // few row tiles (node rows): 128-column blocks give twice the workgroups, i.e. two per CU instead of one
  // (node linears 0.49 -> 0.42 ms/step at 8000 rows x 1024 outputs; 64-column blocks measured slower: 0.475;
  // round 2: two / four k slabs per LDS step - half / a quarter of the barriers - measured 0.433 / 0.546 vs 0.431 ms/step)
  if (k == 256 && !tiled_out && n_out % 128 == 0 && ((m + 127) / 128) * (n_out / 256) < 512) {
    // the node-row shape (wide output, few row blocks): rows register resident, weights streamed through LDS (node_linear.hip)
    if (NODE_LINEAR_DEPTH == 0 && (mode == 1 || mode == 3) && residual == nullptr && ldy == n_out && sc.x_scale == 1.0f &&
        (mode == 1 || (sc.row_scale != nullptr && sc.w_inv != nullptr)))
      return node_linear(x, sc.row_scale, m, wp, plane_stride, mode, n_out, sc.w_inv, bias, y, stream);
    if (NODE_LINEAR_DEPTH != 1) {      // (profiling library: key 8 = 1 restores the one-step lookahead for A/B)
      if (mode == 1) return launch_split<256, 128, 2, Bf16, false, 4>(DIFUSCO_SPLIT_ARGS);
      if (mode == 3) return launch_split<256, 128, 2, Fp16, false, 4>(DIFUSCO_SPLIT_ARGS);
    }
    DIFUSCO_SPLIT_CASE(256, 128)
  }
  DIFUSCO_SPLIT_CASE(256, 256)
  DIFUSCO_SPLIT_CASE(128, 128)
  DIFUSCO_SPLIT_CASE(64, 64)
#undef DIFUSCO_SPLIT_CASE
#undef DIFUSCO_SPLIT_ARGS
  return hipErrorInvalidValue;
}

// Y = ScalarEmbeddingSine(x) W^T + b with the embedding generated inside the kernel (k = n_out = 256 only; modes 1 and 3).
// The generated operand is sin / cos, |.| <= 1: its fp16 planes are scaled by the constant 2^14.
hipError_t linear_scalar_embed_split(const float* x, const int* perm, const float* dimt, const unsigned short* wp,
                                     long long plane_stride, int mode, const float* bias, float* y, long long m,
                                     hipStream_t stream, int tiled_out, const float* w_inv, float* tile_max) {
  if (m <= 0) return hipSuccess;
  if (tile_max != nullptr && !tiled_out) return hipErrorInvalidValue;
  if (mode == 1)
    return launch_split<256, 256, 2, Bf16, true>(x, wp, plane_stride, bias, nullptr, y, m, 256, 256, stream, tiled_out, perm, dimt,
                                                 nullptr, 1.0f, nullptr, tile_max);
  if (mode == 3)
    return launch_split<256, 256, 2, Fp16, true>(x, wp, plane_stride, bias, nullptr, y, m, 256, 256, stream, tiled_out, perm, dimt,
                                                 nullptr, 16384.0f, w_inv, tile_max);
  return hipErrorInvalidValue;
}

// scale[r] = 2^k with max_c |x[r][c]| 2^k in [2^14, 2^15) (k clamped, see pow2_scale_for): the per-row operand scale of the
// fp16 split path.  One wavefront per row.
__global__ __launch_bounds__(256) void row_pow2_scale_kernel(const float* __restrict__ x, long long m, int k,
                                                             float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const long long r = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= m) return;
  float mx = 0.0f;
  for (int c = lane; c < k; c += 64) mx = __builtin_fmaxf(mx, __builtin_fabsf(x[r * k + c]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, off, 64));
  float inv;
  const float sc = pow2_scale_for(mx, inv);
  if (lane == 0) scale[r] = sc;
}

hipError_t launch_row_pow2_scale(const float* x, long long m, int k, float* scale, hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(row_pow2_scale_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, x, m, k, scale);
  return hipGetLastError();
}

}  // namespace difusco
