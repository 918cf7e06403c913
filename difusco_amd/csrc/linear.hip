// fp32 row-linear on the gfx950 matrix cores:  Y[m, :] = X[m, :] * W^T + b (+ residual).
//
// This is the dense block of the DIFUSCO GNN layer: the five nn.Linear(H,H) of GNNLayer
// (difusco/models/gnn_encoder.py:52-56, applied at :94-104), per_layer_out[l][2] (:339-347) and the
// node/edge embedding linears (:303-304).  All of them are [rows, H] x [H, H]^T with H <= 256, i.e. a
// tall-skinny GEMM whose weight fits on chip; rows = E (edges) for the dominant ones.
//
// CDNA4 mapping (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32, 64 cycles/SIMD):
//   * computed TRANSPOSED:  D[f][m] = sum_k W[f][k] * X[m][k]  -> A operand = weight rows (32 output
//     features), B operand = 32 data rows.  In the 32x32 accumulator layout a lane then owns 4
//     CONSECUTIVE output features of ONE data row per register quad, so bias / residual / store are
//     float4 accesses, and (later) a LayerNorm over H is an in-lane sum + one cross-half exchange.
//   * one workgroup = 4 waves = 128 data rows x FB output features; each wave owns 32 data rows and
//     all FB features (FB/32 accumulator blocks of 16 VGPRs).
//   * the K order inside a BK slab is permuted consistently for A and B (lane half hh reads the float4
//     at k = 8q+4hh..+3) so that ONE ds_read_b128 feeds 4 consecutive MFMA k-steps.
//   * LDS rows are padded to BK+4 floats: the 16-lane groups of ds_read_b128 then hit 16 distinct
//     16-byte bank slots (stride 36 or 20 dwords -> row*9 or row*5 mod 16 distinct).
//   * global -> LDS through registers, next slab prefetched while the current one is multiplied.
#include "common.h"
#include "kernels.h"

namespace difusco {

template <int K, int FB, int BK>
__global__ __launch_bounds__(256, 2) void linear_rows_kernel(const float* __restrict__ X,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          const float* residual, float* Y,
                                                          long long M, long long ldy) {
  constexpr int RB = 128;          // data rows per workgroup
  constexpr int NB = FB / 32;      // accumulator blocks per wave
  constexpr int LDS_STRIDE = BK + 4;
  constexpr int C4 = BK / 4;       // float4 per slab row
  constexpr int XV = (RB * C4 + 255) / 256;  // float4 per thread for the X slab
  constexpr int WV = (FB * C4 + 255) / 256;  // float4 per thread for the W slab
  static_assert(K % BK == 0 && BK % 8 == 0, "K must be a multiple of BK, BK of 8");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [RB][LDS_STRIDE]
  float* Ws = smem + RB * LDS_STRIDE;     // [FB][LDS_STRIDE]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hh = lane >> 5;
  const long long r0 = (long long)blockIdx.x * RB;
  const int f0 = blockIdx.y * FB;

  v16f acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  v4f xr[XV], wr[WV];

  // slab loads/stores are written as macros over statically indexed register arrays (a lambda that
  // captures the arrays by reference sends them to scratch memory)
#define DIFUSCO_LOAD_SLAB(KT)                                                                   \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < XV; ++i) {                                            \
      const int idx = tid + 256 * i;                                                            \
      const int row = idx / C4, c4 = idx % C4;                                                  \
      long long gr = r0 + row;                                                                  \
      gr = gr < M ? gr : M - 1; /* rows past M: computed on valid data, never stored */         \
      xr[i] = *reinterpret_cast<const v4f*>(X + gr * K + (KT) + c4 * 4);                     \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < WV; ++i) {                                            \
      int idx = tid + 256 * i;                                                                  \
      if (FB * C4 % 256 != 0) idx = idx < FB * C4 ? idx : FB * C4 - 1;                          \
      const int row = idx / C4, c4 = idx % C4;                                                  \
      wr[i] = *reinterpret_cast<const v4f*>(W + (long long)(f0 + row) * K + (KT) + c4 * 4);  \
    }                                                                                           \
  }
#define DIFUSCO_STORE_SLAB()                                                                    \
  {                                                                                             \
    _Pragma("unroll") for (int i = 0; i < XV; ++i) {                                            \
      const int idx = tid + 256 * i;                                                            \
      const int row = idx / C4, c4 = idx % C4;                                                  \
      *reinterpret_cast<v4f*>(Xs + row * LDS_STRIDE + c4 * 4) = xr[i];                       \
    }                                                                                           \
    _Pragma("unroll") for (int i = 0; i < WV; ++i) {                                            \
      int idx = tid + 256 * i;                                                                  \
      if (FB * C4 % 256 != 0) idx = idx < FB * C4 ? idx : FB * C4 - 1;                          \
      const int row = idx / C4, c4 = idx % C4;                                                  \
      *reinterpret_cast<v4f*>(Ws + row * LDS_STRIDE + c4 * 4) = wr[i];                       \
    }                                                                                           \
  }

  DIFUSCO_LOAD_SLAB(0)
  DIFUSCO_STORE_SLAB()
  __syncthreads();

  const float* xrow = Xs + (wave * 32 + l31) * LDS_STRIDE + hh * 4;
  const float* wrow = Ws + l31 * LDS_STRIDE + hh * 4;

  for (int kt = 0; kt < K; kt += BK) {
    // prefetch the next slab into registers (the last iteration re-reads its own slab: harmless and
    // keeps the register arrays unconditionally defined)
    const int kn = (kt + BK) < K ? kt + BK : kt;
    DIFUSCO_LOAD_SLAB(kn)
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      const float4 xb = *reinterpret_cast<const float4*>(xrow + q * 8);
      float4 wa[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        wa[nb] = *reinterpret_cast<const float4*>(wrow + nb * 32 * LDS_STRIDE + q * 8);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nb].x, xb.x, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nb].y, xb.y, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nb].z, xb.z, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nb].w, xb.w, acc[nb], 0, 0, 0);
    }
    __syncthreads();
    DIFUSCO_STORE_SLAB()
    __syncthreads();
  }
#undef DIFUSCO_LOAD_SLAB
#undef DIFUSCO_STORE_SLAB

  // Epilogue.  Accumulator register r of block nb, lane (l31, hh):
  //   feature = f0 + nb*32 + (r&3) + 8*(r>>2) + 4*hh ,  data row = r0 + wave*32 + l31.
  const long long row = r0 + wave * 32 + l31;
  if (row < M) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = f0 + nb * 32 + 8 * g + 4 * hh;
        float4 v = make_float4(acc[nb][4 * g + 0], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]);
        if (bias != nullptr) {
          const float4 b = *reinterpret_cast<const float4*>(bias + f);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (residual != nullptr) {
          const float4 rr = *reinterpret_cast<const float4*>(residual + row * ldy + f);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4*>(Y + row * ldy + f) = v;
      }
    }
  }
}

template <int K, int FB, int BK>
static hipError_t launch_linear(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                long long m, int n_out, long long ldy, hipStream_t stream) {
  constexpr size_t lds = (size_t)(128 + FB) * (BK + 4) * sizeof(float);
  static std::atomic<unsigned long long> attr_devices{0};
  {
    hipError_t e = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&linear_rows_kernel<K, FB, BK>), (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((m + 127) / 128), (unsigned)(n_out / FB));
  hipLaunchKernelGGL((linear_rows_kernel<K, FB, BK>), grid, dim3(256), lds, stream, x, w, bias, residual, y, m, ldy);
  return hipGetLastError();
}

// n_out must be a multiple of the feature block: 256 when n_out % 256 == 0, else 128 / 64 / 32.
hipError_t linear_rows(const float* x, const float* w, const float* bias, const float* residual, float* y,
                       long long m, int k, int n_out, long long ldy, hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  // Feature block: the widest one that still gives the chip a workgroup per CU (node rows: 8,000 x 256 outputs are 63 row
  // blocks - with 256-wide blocks 63 workgroups on 256 CUs, 52 us; with 64-wide blocks 252).  Every output element is the
  // same k-ordered fma chain whatever the block width: results are bit-identical.
  const long long rb = (m + 127) / 128;
  const bool wide = n_out % 256 == 0 && rb * (n_out / 256) >= 256;
  const bool mid = n_out % 128 == 0 && (n_out % 64 != 0 || rb * (n_out / 128) >= 256);
#define DIFUSCO_LIN_CASE(KK, BKK)                                                                         \
  if (k == KK) {                                                                                          \
    if (wide) return launch_linear<KK, 256, BKK>(x, w, bias, residual, y, m, n_out, ldy, stream);         \
    if (mid) return launch_linear<KK, 128, BKK>(x, w, bias, residual, y, m, n_out, ldy, stream);          \
    if (n_out % 64 == 0) return launch_linear<KK, 64, BKK>(x, w, bias, residual, y, m, n_out, ldy, stream);   \
    if (n_out % 32 == 0) return launch_linear<KK, 32, BKK>(x, w, bias, residual, y, m, n_out, ldy, stream);   \
    return hipErrorInvalidValue;                                                                          \
  }
  DIFUSCO_LIN_CASE(256, 16)
  DIFUSCO_LIN_CASE(128, 16)
  DIFUSCO_LIN_CASE(64, 16)
  DIFUSCO_LIN_CASE(32, 16)
#undef DIFUSCO_LIN_CASE
  return hipErrorInvalidValue;
}

}  // namespace difusco
