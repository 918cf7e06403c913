// Edge embedding of a step whose edge input is NOT a two-row table (Gaussian diffusion; categorical x_t that is not exactly {0,1}):
//   e0[s] = edge_embed( ScalarEmbeddingSine(x_t[s]) )        gnn_encoder.py:230-249 (features), :304,:395 (the linear)
// for gfx950, H = 256, written straight into the TILED edge state the fused layers read (kernels.h: edge_tiled_offset) together with
// the per-tile max |e| that scales the next kernel's fp16 operand.
//
// Same dataflow as GEMM 1 of the fused edge kernel (edge_layer_kernel.h): D[f][edge] = sum_k W[f][k] X[edge][k] with the weight planes
// as the A operand, streamed through LDS in 16 KiB stages ([256 rows][16 k] x 2 planes, double buffered, LDS-DMA, XOR-swizzled rows,
// one barrier per stage), and the 32 edges of a wave as the B operand - here GENERATED: the sixteen k values a lane contributes to
// slab t are four (sin, cos) pairs of x / dim_t, one precise sincosf each, computed between the request of stage t + 1 and the MFMAs of
// stage t (the vector work covers the DMA latency).  A lane ends with the 128 features {32 nb + 8 g + 4 hh + 0..3} of ITS edge: float4
// stores of whole KiB per wave instruction, no LDS round trip.
// The general row-linear with generated rows (linear_split.hip, GEN) stages both operands through LDS with two barriers per 16-k step:
// 0.78 ms per launch at E = 10^6; this kernel is what the step driver uses on the fused path (round 5), the general one stays for
// the unfused sequence.  Arithmetic of the features: that of scalar_embed_kernel / the GEN path (x / dim_t[c], precise sincosf).
//
// ROUND 6 - the TABLE kernel (edge_embed_table_kernel below).  e0 is a function of ONE scalar per edge: e0(x) = W_emb [sin(x w_k),
// cos(x w_k)]_k + b_emb, a smooth curve in R^256 (128 frequencies w_k = 10^(-4k/128) <= 1).  `gen_table` holds that curve sampled at
// x_r = kGenXMin + (r - 3) / 4 (71 rows of 1 KiB, built once per weight blob by difusco_gen_table_build with the EXACT fp32 kernels:
// scalar_embed_kernel + the fp32-MFMA row linear).  A 32-edge tile whose edges all lie in [-8, 8) is evaluated by EIGHT-point (degree 7)
// Lagrange interpolation of the rows around each edge: no sincosf, no MFMA.  Interpolation error on this grid: 2.3e-9 (measured against
// float64 rows, scratch check in tests/test_gpu_round6.py), far below the fp32 rounding of the rows themselves (1.3e-6: the reference's own
// fp32 distance from float64 is 1.7e-6); the position inside the cell, (x - x_i) * 4, is EXACT in fp32 (grid points are multiples of 2^-2).
// A tile with an edge outside the table (|x| >= 8: never reached by x_t of a diffusion run, but the ABI takes any float) or a non-finite x
// is FLAGGED and left to the contraction kernel, which then runs only the workgroups that hold a flagged tile.
#include "edge_layer_common.h"

namespace difusco {

namespace edge_embed {
constexpr int H = 256, WAVES = 4, NS = 16;      // 16 stages = 16 k slabs
constexpr int PLANE = 256 * 16;                 // 16-bit elements of one plane of one stage
constexpr int BUF = 2 * PLANE;                  // one stage: 2 planes (16 KiB)
}  // namespace edge_embed

template <typename T>
__global__ __launch_bounds__(256, 2) void edge_embed_tiled_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                                  const float* __restrict__ dimt,
                                                                  const unsigned short* __restrict__ planes, long long plane_stride,
                                                                  const float* __restrict__ w_inv, const float* __restrict__ bias,
                                                                  float* __restrict__ e, long long n_edges,
                                                                  float* __restrict__ tile_max,
                                                                  const int* __restrict__ tile_flag) {
  using namespace edge_embed;
  typedef typename T::frag frag;
  // (ONE __shared__ object: the two stage buffers and, behind them, the 256 dim_t values.  dim_t is read from LDS, not from global
  //  memory: a global load issued after a stage request sits behind the LDS-DMA pieces in the in-order vmcnt queue, so waiting for
  //  it would wait for the stage - the double buffering would be gone)
  __shared__ __attribute__((aligned(16))) unsigned short wbuf[2 * BUF + 2 * H];      // 32 KiB + 1 KiB
  float* const dimt_s = reinterpret_cast<float*>(wbuf + 2 * BUF);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long tile = (long long)blockIdx.x * WAVES + wave;
  const long long s_raw = tile * 32 + l31;
  const bool valid = s_raw < n_edges;
  const long long s = valid ? s_raw : n_edges - 1;      // lanes past the end redo the last edge and store nothing

  // ---- weight stages by LDS-DMA (the fused kernel's scheme for GEMM 1: wave w moves pieces 2 w, 2 w + 1 of the eight 1-KiB pieces of
  // a plane; lane L fills slot (piece, L) = entry piece * 32 + L / 2, and fetches the half that belongs there - wslot's XOR on the source)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(planes), 0, 0x7fffffff, 0x00020000);
  const int plane_bytes = (int)plane_stride * 2;
  unsigned dvoff;
  {
    const int entry0 = (2 * wave) * 32 + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
    dvoff = (unsigned)(entry0 * 16 + half * 8) * 2;      // bytes inside the stage's [256 rows][16 k] slab
  }
#define EMBED_DMA_STAGE(t)                                                                                          \
  {                                                                                                                 \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w,                                                                \
          (__attribute__((address_space(3))) void*)(wbuf + ((t) & 1) * BUF + pl * PLANE + (2 * wave + i) * 512), 16, \
          dvoff, (t) * 8192 + pl * plane_bytes + i * 1024, 0, 0);                                                    \
  }
  // tile_flag (round 6): the table kernel has already written every tile it could; it flagged the others.  A workgroup runs when one
  // of its four tiles is flagged (it then recomputes all four: the weight-stage barriers are workgroup wide).
  if (tile_flag != nullptr) {
    if (!__syncthreads_or(tile_flag[tile] != 0)) return;
  }
  const float xv = x[perm ? perm[s] : s];
  dimt_s[tid] = dimt[tid];      // (256 threads, 256 features; visible after the barrier below)
  EMBED_DMA_STAGE(0)

  constexpr float kXScale = T::kScaled ? 16384.0f : 1.0f;      // |sin|, |cos| <= 1: the fp16 planes are those of the features x 2^14
  v16f acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces of stage 0
  __syncthreads();
  const int a_off = wslot(l31, hh);
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    if (t + 1 < NS) {      // into the other buffer: every wave left it at the last barrier
      EMBED_DMA_STAGE(t + 1)
      __builtin_amdgcn_sched_barrier(0);
    }
    // B operand of slab t: features 16 t + {4 hh .. 4 hh + 3} and 16 t + 8 + {4 hh .. 4 hh + 3} of this lane's edge = four (sin, cos)
    // pairs (features 2 j, 2 j + 1 share dim_t)
    frag xh, xl;
    {
      float xs[8];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int c = 16 * t + 8 * (p >> 1) + 4 * hh + 2 * (p & 1);
        const float v = xv / dimt_s[c];
        float sn, cs;
        sincosf(v, &sn, &cs);
        xs[2 * p] = sn * kXScale;
        xs[2 * p + 1] = cs * kXScale;
      }
      split8<T>(xs, xh, xl);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned short* wb = wbuf + (t & 1) * BUF + a_off;
    frag fh[3], fl[3];
#define EMBED_FRAG(bi, slot)                                                       \
  {                                                                                \
    fh[slot] = *reinterpret_cast<const frag*>(wb + (bi) * 32 * 16);                \
    fl[slot] = *reinterpret_cast<const frag*>(wb + PLANE + (bi) * 32 * 16);        \
  }
    EMBED_FRAG(0, 0)
    EMBED_FRAG(1, 1)
#pragma unroll
    for (int bi = 0; bi < 8; ++bi) {
      if (bi + 2 < 8) EMBED_FRAG(bi + 2, (bi + 2) % 3)
      __builtin_amdgcn_sched_barrier(0);
      acc[bi] = T::mfma(fl[bi % 3], xh, acc[bi]);      // smallest terms first
      acc[bi] = T::mfma(fh[bi % 3], xl, acc[bi]);
      acc[bi] = T::mfma(fh[bi % 3], xh, acc[bi]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef EMBED_FRAG
    if (t + 1 < NS) {
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): stage t + 1 has landed (this wave's pieces)
      __syncthreads();
    }
  }
#undef EMBED_DMA_STAGE

  // ---- epilogue: undo the operand scales, add the bias, store the tile, leave its max |e| for the next kernel -------------------
  float* const etile = e + tile * (32 * H);
  constexpr float kInvX = T::kScaled ? 1.0f / 16384.0f : 1.0f;
  float tmx = 0.0f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * nb + 8 * g + 4 * hh;
      const v4f b = *reinterpret_cast<const v4f*>(bias + f);
      v4f sc = {kInvX, kInvX, kInvX, kInvX};
      if (w_inv != nullptr) sc = *reinterpret_cast<const v4f*>(w_inv + f) * kInvX;      // (powers of two: exact)
      v4f v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[nb][4 * g + q] * sc[q] + b[q];
      if (valid) {
        *reinterpret_cast<v4f*>(etile + (2 * nb + (g >> 1)) * 512 + (g & 1) * 256 + lane * 4) = v;
        tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
        tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
      }
    }
  if (tile_max != nullptr) {      // (pad lanes contribute 0; a tile entirely past the end - the padding of the last workgroup - reports 0)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) tmx = __builtin_fmaxf(tmx, __shfl_xor(tmx, off, 64));
    if (lane == 0) tile_max[tile] = tmx;
  }
}

// ---- the table kernel ---------------------------------------------------------------------------------------------------------
// One persistent workgroup of 8 waves per CU; LDS = the 71 table rows (71 KiB) + a transpose buffer of 8 edges x 1 KiB per wave.
// A wave takes a 32-edge tile: lanes 0..31 hold the edges' cell index and their eight Lagrange weights; for each edge (its index and
// weights broadcast by v_readlane) lane L accumulates the features 4 L .. 4 L + 3 from the eight rows - all lanes read the SAME row at
// consecutive 16-byte slots: conflict free, one KiB per ds_read_b128 - and eight edges at a time go through the transpose buffer (row
// pitch 66 slots: both directions conflict free) so that a store instruction writes eight full 128-byte lines of the tiled layout.
namespace gen {
constexpr int NPT = 8, WAVES = 8;
constexpr int EB = 2;                                 // edges whose row reads are in flight together (1: 261.7 us, 2: 216.5 us, 4: 218.2 us at E = 10^6)
constexpr int TAB_FLOATS = kGenRows * 256;
constexpr int PITCH = 66 * 4;                         // floats per staged edge row
constexpr int LDS_BYTES = (TAB_FLOATS + WAVES * 8 * PITCH) * 4;
}  // namespace gen

__global__ __launch_bounds__(512, 1) void edge_embed_table_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                                  const float* __restrict__ table, float* __restrict__ e,
                                                                  long long n_edges, float* __restrict__ tile_max,
                                                                  int* __restrict__ tile_flag) {
  using namespace gen;
  extern __shared__ __attribute__((aligned(16))) float smem_gen[];
  float* const tab = smem_gen;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const stage = smem_gen + TAB_FLOATS + wave * (8 * PITCH);
  for (int i = tid; i < TAB_FLOATS / 4; i += 512)
    *reinterpret_cast<v4f*>(tab + 4 * i) = *reinterpret_cast<const v4f*>(table + 4 * i);
  __syncthreads();
  const long long n_tiles = (n_edges + 31) / 32, n_tiles_pad = (n_edges + 127) / 128 * 4;      // (pad tiles of the last 128-edge group report max 0)
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles_pad; tile += (long long)gridDim.x * WAVES) {
    if (tile >= n_tiles) {
      if (lane == 0 && tile_max != nullptr) tile_max[tile] = 0.0f;
      continue;
    }
    const long long s_raw = tile * 32 + l31;
    const bool valid = s_raw < n_edges;
    const long long s = valid ? s_raw : n_edges - 1;
    const float xv = x[perm ? perm[s] : s];
    const float u = (xv - kGenXMin) * kGenInvH;
    const bool inside = u >= 0.0f && u < (float)kGenIntervals;      // (false for NaN)
    if (__ballot(!inside) != 0ull) {      // (wave uniform) leave the tile to the contraction kernel
      if (lane == 0) tile_flag[tile] = 1;
      continue;
    }
    int ci = (int)u;      // cell: rows ci .. ci + 7 hold the grid points x_c - 3h .. x_c + 4h, x_c = kGenXMin + ci h
    const float sp = (xv - (kGenXMin + (float)ci * kGenH)) * kGenInvH;      // position in the cell, exact (Sterbenz; powers of two)
    // Lagrange weights of the nodes -3 .. 4 at sp: w_a = prod_{b != a} (sp - b) / (a - b)
    float w[NPT];
    {
      const float d[NPT] = {sp + 3.0f, sp + 2.0f, sp + 1.0f, sp, sp - 1.0f, sp - 2.0f, sp - 3.0f, sp - 4.0f};
      // 1 / prod_{b != a} (a - b) for a = -3 .. 4:  -1/5040, 1/720, -1/240, 1/144, -1/144, 1/240, -1/720, 1/5040
      constexpr float c[NPT] = {-1.0f / 5040.0f, 1.0f / 720.0f, -1.0f / 240.0f, 1.0f / 144.0f, -1.0f / 144.0f, 1.0f / 240.0f, -1.0f / 720.0f, 1.0f / 5040.0f};
#pragma unroll
      for (int a = 0; a < NPT; ++a) {
        float pr = c[a];
#pragma unroll
        for (int b = 0; b < NPT; ++b)
          if (b != a) pr *= d[b];
        w[a] = pr;
      }
    }
    float* const etile = e + tile * (32 * 256);
    float tmx = 0.0f;
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
      for (int ed = 0; ed < 8; ed += EB) {      // EB edges per step: their 8 EB row reads are issued before the first multiply-add
        constexpr int order[NPT] = {0, 7, 1, 6, 2, 5, 3, 4};      // smallest weights first: the outer nodes, then inwards
        v4f rr[EB][NPT];
#pragma unroll
        for (int k2 = 0; k2 < EB; ++k2) {
          const int row = __builtin_amdgcn_readlane(ci, 8 * grp + ed + k2);
          const float* r = tab + row * 256 + lane * 4;
#pragma unroll
          for (int j = 0; j < NPT; ++j) rr[k2][j] = *reinterpret_cast<const v4f*>(r + j * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < EB; ++k2) {
          const int edge = 8 * grp + ed + k2;
          v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int jj = 0; jj < NPT; ++jj) {
            const int j = order[jj];
            const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w[j]), edge));
            acc += rr[k2][j] * wj;
          }
          *reinterpret_cast<v4f*>(stage + (ed + k2) * PITCH + lane * 4) = acc;
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) {      // lane L: edge 8 grp + (L & 7), feature quad Q = 8 j + (L >> 3)
        const int q = 8 * j + (lane >> 3), sidx = 8 * grp + (lane & 7);
        const v4f v = *reinterpret_cast<const v4f*>(stage + (lane & 7) * PITCH + q * 4);
        if (tile * 32 + sidx < n_edges) {
          __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(etile + (q >> 2) * 512 + ((q >> 1) & 1) * 256 + ((q & 1) * 32 + sidx) * 4));
          tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
          tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (tile_max != nullptr) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) tmx = __builtin_fmaxf(tmx, __shfl_xor(tmx, off, 64));
      if (lane == 0) tile_max[tile] = tmx;
    }
  }
}

// mode: 1 = bf16 planes (unscaled), 3 = fp16 planes (w_inv required).  planes: first plane of that type, [16 slabs][256 rows][16];
// e: tiled [ceil(n_edges / 256) * 256, 256]; tile_max: one float per 32-edge tile of the padded range, or null.
// gen_table + tile_flag (both or neither; tile_flag: ceil(n_edges / 128) * 4 ints of scratch): the table kernel writes every tile whose
// edges lie inside the table, flags the others, and the contraction kernel then runs only the workgroups with a flagged tile.
hipError_t launch_edge_embed_tiled(const float* x, const int* perm, const float* dimt, const unsigned short* planes, long long plane_stride,
                                   int mode, const float* w_inv, const float* bias, float* e, long long n_edges, float* tile_max,
                                   hipStream_t stream, const float* gen_table, int* tile_flag) {
  if (n_edges <= 0) return hipSuccess;
  if ((mode != 1 && mode != 3) || (mode == 3 && w_inv == nullptr) || ((gen_table == nullptr) != (tile_flag == nullptr))) return hipErrorInvalidValue;
  const unsigned grid = (unsigned)((n_edges + 127) / 128);
  if (gen_table != nullptr) {
    static std::atomic<unsigned long long> attr_devices{0};
    hipError_t er = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&edge_embed_table_kernel), 160 * 1024);
    if (er != hipSuccess) return er;
    er = hipMemsetAsync(tile_flag, 0, sizeof(int) * (size_t)grid * 4, stream);
    if (er != hipSuccess) return er;
    static std::atomic<int> n_cu{0};
    int cus = n_cu.load(std::memory_order_relaxed);
    if (cus == 0) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
      n_cu.store(cus, std::memory_order_relaxed);
    }
    const long long n_tiles_pad = (long long)grid * 4;
    unsigned tgrid = (unsigned)((n_tiles_pad + gen::WAVES - 1) / gen::WAVES);
    if (tgrid > (unsigned)cus) tgrid = (unsigned)cus;
    hipLaunchKernelGGL(edge_embed_table_kernel, dim3(tgrid), dim3(512), gen::LDS_BYTES, stream, x, perm, gen_table, e, n_edges, tile_max, tile_flag);
    er = hipGetLastError();
    if (er != hipSuccess) return er;
  }
  if (mode == 3)
    hipLaunchKernelGGL((edge_embed_tiled_kernel<FFp16>), dim3(grid), dim3(256), 0, stream, x, perm, dimt, planes, plane_stride, w_inv, bias, e,
                       n_edges, tile_max, tile_flag);
  else
    hipLaunchKernelGGL((edge_embed_tiled_kernel<FBf16>), dim3(grid), dim3(256), 0, stream, x, perm, dimt, planes, plane_stride, nullptr, bias, e,
                       n_edges, tile_max, tile_flag);
  return hipGetLastError();
}

// x[r] = kGenXMin + (r - 3) h, r < kGenRows: the sample points of the table (exact: multiples of 2^-2)
__global__ void gen_table_grid_kernel(float* __restrict__ x) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < kGenRows) x[r] = kGenXMin + (float)(r - 3) * kGenH;
}

hipError_t launch_gen_table_grid(float* x, hipStream_t stream) {
  hipLaunchKernelGGL(gen_table_grid_kernel, dim3((kGenRows + 255) / 256), dim3(256), 0, stream, x);
  return hipGetLastError();
}

}  // namespace difusco
