// Node-row linear of a DIFUSCO layer for gfx950: node4 = h W4^T + b with W4 = [U | V | A | B] (gnn_encoder.py:52-55,94-103), the
// [N, 256] x [256, 1024] product that feeds the fused edge kernel's neighbour tables, on the split-precision matrix-core path.
//
// Same dataflow as GEMM 1 of the fused edge kernel (edge_layer_kernel.h): the DATA rows are a register-resident MFMA operand, the
// WEIGHTS stream through LDS.
//   * a wave owns 32 node rows: their 256 features are loaded once as full lines, transposed through LDS to lane = row, scaled by
//     the row's power-of-two operand scale and split into two 16-bit planes in registers (128 VGPRs) - no per-k-step barrier pair;
//   * a workgroup (4 waves, 128 rows) computes 128 output columns: the weight planes of those columns stream through LDS in 8
//     stages of 2 slabs x [128 rows][16 k] x 2 planes (16 KB) by LDS-DMA, three buffers, requests two stages ahead of the MFMAs
//     (counted vmcnt waits), one barrier per stage; rows XOR-swizzled like the fused kernel's stages (conflict-free ds_read_b128);
//   * D[row][f] accumulators: a lane owns one output feature of 16 rows per block, a store instruction writes 2 rows x 128 B.
// The general row-linear (linear_split.hip) stages both operands through LDS with two barriers per 16-k step.  At 8,000 rows the
// whole product is ONE round of 504 workgroups, so the load, MFMA and store phases of all CUs coincide and add up (timing ablations:
// scripts/bench_node_linear.py, profiles/r03/node_linear_microbench.json); what shortened it was the full-line X loads and the
// full-line stores, not the leaner stage loop: rocprofv3 29.3 -> 22.9 us per launch.
#include "edge_layer_common.h"

namespace difusco {

namespace nodelin {
constexpr int K = 256, RB = 128, CB = 128, NSLAB = K / 16;
constexpr int SPS = 2;                       // k slabs per stage
constexpr int NSTAGE = NSLAB / SPS;          // 8 stages of 24 MFMAs per wave
constexpr int NBUF = 3;                      // stage buffers: requests run two stages ahead of the MFMAs
constexpr int PLANE = CB * 16;               // 16-bit elements of one plane of one slab (4 KB)
constexpr int BUF = SPS * 2 * PLANE;         // one stage: SPS slabs x 2 planes (16 KB)
}  // namespace nodelin

// ABL (profiling library only, timing ablations - results are wrong): bit 0 no output stores, bit 1 no X loads, bit 2 no weight
// stream / MFMA loop; 16 = direct lane = row loads of X (correct results, the first version of this kernel)
template <typename T, int ABL = 0>
__global__ __launch_bounds__(256, 2) void node_linear_kernel(const float* __restrict__ X, long long M,
                                                             const float* __restrict__ row_scale,
                                                             const unsigned short* __restrict__ planes, long long plane_stride,
                                                             int n_out, const float* __restrict__ w_inv,
                                                             const float* __restrict__ bias, float* __restrict__ Y) {
  using namespace nodelin;
  typedef typename T::frag frag;
  // (keep this the ONLY __shared__ object of the kernel: with a second one the compiler's LDS-DMA tracking put an
  // s_waitcnt vmcnt(0) in front of the first ds_read after every stage request)
  __shared__ __attribute__((aligned(16))) unsigned short wbuf[NBUF * BUF];   // 48 KB
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid: the column blocks of one row block run back to back on ONE XCD (workgroup b -> XCD b % 8), so the X rows are
  // fetched into one L2 once
  const int ncb = n_out / CB, slot = (int)blockIdx.x >> 3;
  const int rb_i = (slot / ncb) * 8 + ((int)blockIdx.x & 7), cb_i = slot % ncb;
  if ((long long)rb_i * RB >= M) return;
  const long long row_raw = (long long)rb_i * RB + wave * 32 + l31;
  const long long row = row_raw < M ? row_raw : M - 1;
  const int f0 = cb_i * CB;

  // ---- weight stages by LDS-DMA: stage t = slabs SPS t .., rows f0 .. f0 + 127 of both planes; a wave moves its 32 rows -------
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(planes), 0, 0x7fffffff, 0x00020000);
  const int plane_bytes = (int)plane_stride * 2;
  unsigned dvoff;
  {
    const int entry = wave * 32 + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
    dvoff = (unsigned)((f0 + entry) * 32 + half * 16);
  }
#define NODELIN_DMA_STAGE(t)                                                                                            \
  {                                                                                                                     \
    _Pragma("unroll") for (int sl = 0; sl < SPS; ++sl)                                                                  \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w,                                                                    \
          (__attribute__((address_space(3))) void*)(wbuf + ((t) % NBUF) * BUF + (sl * 2 + pl) * PLANE + wave * 512), 16, \
          dvoff, (SPS * (t) + sl) * n_out * 32 + pl * plane_bytes, 0, 0);                                               \
  }
  if constexpr (!(ABL & 4)) {
    NODELIN_DMA_STAGE(0)
    NODELIN_DMA_STAGE(1)
  }

  // ---- the wave's 32 rows -> scaled 16-bit planes in registers -----------------------------------------------------------------
  // The MFMA operand wants lane = row, but a load with lane = row touches 32 lines for 1 KB (measured: 3 - 4 us of the kernel).
  // Rows are therefore loaded as FULL LINES - unit u = the 128 B [32 u, 32 u + 32) of the 32 rows, four instructions of 8 rows x
  // 8 chunks each, all 32 issued up front - and transposed through the wave's 4 KB share of stage buffer 2, which is idle until
  // the first barrier (chunk position XOR-swizzled like the fused kernel's full-line gathers: stores and reads conflict free).
  const float sc = row_scale != nullptr ? row_scale[row] : 1.0f;
  frag xh[NSLAB], xl[NSLAB];
  if constexpr (!(ABL & 16) && !(ABL & 2)) {
    const long long rbase = (long long)rb_i * RB + wave * 32;
    v4f stage[8][4];
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
      const long long r = rbase + 8 * p4 + (lane >> 3);
      const float* src = X + (r < M ? r : M - 1) * K + 4 * (lane & 7);
#pragma unroll
      for (int u = 0; u < 8; ++u) stage[u][p4] = *reinterpret_cast<const v4f*>(src + 32 * u);
    }
    char* area = reinterpret_cast<char*>(wbuf + 2 * BUF) + wave * 4096;
    int wr_off[4];
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
      const int r = 8 * p4 + (lane >> 3);
      wr_off[p4] = r * 128 + (((lane & 7) ^ (((r >> 1) & 3) | (((r >> 4) & 1) << 2))) << 4);
    }
    const int rswz = ((l31 >> 1) & 3) | (((l31 >> 4) & 1) << 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int p4 = 0; p4 < 4; ++p4) *reinterpret_cast<v4f*>(area + wr_off[p4]) = stage[u][p4];
      v4f c[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) c[g] = *reinterpret_cast<const v4f*>(area + l31 * 128 + (((2 * g + hh) ^ rswz) << 4));
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {      // slab 2u + s2: floats {4hh..4hh+3} and {8+4hh..} of its 16 = chunks 4 s2 + hh, 4 s2 + 2 + hh
        v4f c0 = c[2 * s2], c1 = c[2 * s2 + 1];
        if constexpr (T::kScaled) {
          c0 = c0 * sc;
          c1 = c1 * sc;
        }
        const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        split8<T>(xs, xh[2 * u + s2], xl[2 * u + s2]);
      }
    }
  } else {
    const float* xr = X + row * K + 4 * hh;
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) {
      v4f c0, c1;
      if constexpr (ABL & 2) {
        c0 = v4f{sc, sc + ks, sc * lane, 1.0f};
        c1 = c0 + 1.0f;
      } else {                    // ABL bit 4 (value 16): the direct lane = row loads (the first version of this kernel), for A/B
        c0 = *reinterpret_cast<const v4f*>(xr + 16 * ks);
        c1 = *reinterpret_cast<const v4f*>(xr + 16 * ks + 8);
      }
      if constexpr (T::kScaled) {
        c0 = c0 * sc;
        c1 = c1 * sc;
      }
      const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      split8<T>(xs, xh[ks], xl[ks]);
    }
  }
  v16f acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  // the X loads above were waited for by the split: of the vector-memory queue only stage requests can be pending from here on,
  // and there are no stores before the epilogue - counted waits are safe
  // (__syncthreads() would add a workgroup fence, and the fence waits for EVERY LDS-DMA request: raw s_barrier instead; the asm
  // memory clobber keeps the LDS reads / DMA requests on their side of it)
  static_assert(SPS == 2, "the counted waits below allow one stage = 4 requests per wave in flight");
#define NODELIN_SYNC(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
  NODELIN_SYNC(4);   // stage 0 has landed (stage 1 may be in flight)
  const int a_off = wslot(l31, hh);
#pragma unroll
  for (int t = 0; t < ((ABL & 4) ? 0 : NSTAGE); ++t) {
    if (t + 2 < NSTAGE) {      // into the buffer of stage t - 1, which every wave left at the barrier that ended it
      NODELIN_DMA_STAGE(t + 2)
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int sl = 0; sl < SPS; ++sl) {
      const unsigned short* wb = wbuf + (t % NBUF) * BUF + sl * 2 * PLANE + a_off;
      const int ks = SPS * t + sl;
      frag fh[4], fl[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        fh[nb] = *reinterpret_cast<const frag*>(wb + nb * 32 * 16);
        fl[nb] = *reinterpret_cast<const frag*>(wb + PLANE + nb * 32 * 16);
      }
      // smallest terms first per accumulator; the four accumulator chains alternate
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = T::mfma(xh[ks], fl[nb], acc[nb]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = T::mfma(xl[ks], fh[nb], acc[nb]);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[nb] = T::mfma(xh[ks], fh[nb], acc[nb]);
    }
    if (t + 1 < NSTAGE) {      // stage t + 1 must have landed; stage t + 2 (just requested) may stay in flight
      if (t + 2 < NSTAGE) NODELIN_SYNC(4);
      else NODELIN_SYNC(0);
    }
  }
#undef NODELIN_SYNC
#undef NODELIN_DMA_STAGE

  if constexpr (ABL & 4) {
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) acc[ks & 3][ks] += (float)xh[ks][0] + (float)xl[ks][1] + (float)xh[ks][7] + (float)xl[ks][4];
  }
  // (one 32-column block at a time over all of k, its stores issued inside the stage loop, measured 1 us SLOWER than storing
  // everything here - scripts/bench_node_linear.py)
  // D[row][feature] (rows = the A operand): lane = feature f0 + 32 nb + l31, register r = row 8 (r >> 2) + 4 hh + (r & 3) of the
  // wave's 32, so one store instruction writes 2 rows x 128 B - whole lines.  (With D[feature][row], 16-byte stores of 32 rows x 32 B
  // per instruction, the write counter showed 47.6 MB per launch for the 32.8 MB of node4 and the kernel took 3.7 us longer.)
  // Scales and biases are loaded before the first store: a load issued after a store is waited for through the same counter
  // (vmcnt), i.e. together with the store's acknowledgement.
  const long long rbase = (long long)rb_i * RB + wave * 32;
  if (rbase < M && (!(ABL & 1) || sc == 12345.678f)) {
    float wv[4], bv[4], rs[16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      wv[nb] = w_inv != nullptr ? w_inv[f0 + nb * 32 + l31] : 1.0f;      // (fp16 planes: 2^-k_f; fused path: times log2(e) on the A | B columns)
      bv[nb] = bias != nullptr ? bias[f0 + nb * 32 + l31] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long rr = rbase + 8 * (r >> 2) + 4 * hh + (r & 3);
      rs[r] = 1.0f;     // 1 / 2^k of the row, exact
      if constexpr (T::kScaled) rs[r] = __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(unsigned, row_scale[rr < M ? rr : M - 1]));
    }
    asm volatile("" ::: "memory");
    float* yb = Y + (rbase + 4 * hh) * n_out + f0 + l31;
    const bool full = rbase + 32 <= M;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 8 * (r >> 2) + (r & 3);
      if (full || rbase + 4 * hh + m < M) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) yb[(long long)m * n_out + nb * 32] = acc[nb][r] * (wv[nb] * rs[r]) + bv[nb];
      }
    }
  }
}

#ifdef DIFUSCO_PROFILING
int g_node_linear_ablate = 0;    // difusco_debug_set key 10
#endif

// mode: 1 = bf16 planes (unscaled), 3 = fp16 planes (row_scale / w_inv required); planes = first plane of that type, [K/16][n_out][16]
hipError_t node_linear(const float* x, const float* row_scale, long long m, const unsigned short* planes, long long plane_stride,
                       int mode, int n_out, const float* w_inv, const float* bias, float* y, hipStream_t stream) {
  using namespace nodelin;
  if (m <= 0) return hipSuccess;
  if (n_out % CB != 0 || (mode != 1 && mode != 3) || (mode == 3 && (row_scale == nullptr || w_inv == nullptr))) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)(8 * (((m + RB - 1) / RB + 7) / 8) * (n_out / CB));
#ifdef DIFUSCO_PROFILING
#define NODELIN_ABL_CASE(A)                                                                                                        \
  if (g_node_linear_ablate == A) {                                                                                                 \
    hipLaunchKernelGGL((node_linear_kernel<FFp16, A>), dim3(grid), dim3(256), 0, stream, x, m, row_scale, planes, plane_stride,    \
                       n_out, w_inv, bias, y);                                                                                     \
    return hipGetLastError();                                                                                                      \
  }
  if (mode == 3) {
    NODELIN_ABL_CASE(1) NODELIN_ABL_CASE(2) NODELIN_ABL_CASE(3) NODELIN_ABL_CASE(4) NODELIN_ABL_CASE(5) NODELIN_ABL_CASE(7) NODELIN_ABL_CASE(16)
  }
#undef NODELIN_ABL_CASE
#endif
  if (mode == 3)
    hipLaunchKernelGGL((node_linear_kernel<FFp16>), dim3(grid), dim3(256), 0, stream, x, m, row_scale, planes, plane_stride, n_out, w_inv,
                       bias, y);
  else
    hipLaunchKernelGGL((node_linear_kernel<FBf16>), dim3(grid), dim3(256), 0, stream, x, m, row_scale, planes, plane_stride, n_out, w_inv,
                       bias, y);
  return hipGetLastError();
}

}  // namespace difusco
