// Software-pipelined, persistent variant of the fused edge-layer kernel (same arithmetic, same tile
// decomposition and the same result bits as edge_layer.hip: see that file for the math and the layouts).
//
// Why.  In edge_layer.hip a wave walks GEMM 1 -> epilogue -> GEMM 2 serially; the matrix pipe idles during the
// VALU / address-unit heavy epilogue and the VALU idles during the GEMMs (measured: MFMA busy 38 % of SIMD
// time, phases 26 % / 37 % / 31 %).  Here ONE wave per SIMD (512 registers) runs a two-deep pipeline over its
// tiles, so that the gather / gate / neighbour-sum half of the epilogue of tile k rides under GEMM 2 of tile k-1:
//
//     body(kk):  stages 0..7   GEMM 2 of tile kk-1   |  epilogue part 1 of tile kk (gathers, gate, neighbour sum)
//                              epilogue part 2 of tile kk (LayerNorms, SiLU, split into 16-bit planes; VALU only)
//                stages 8..15  GEMM 1 of tile kk+1
//
// (A three-deep version that also hides part 2 under GEMM 1 needs two accumulator sets + planes = more than the
// 256 VALU-addressable registers of a wave.)  Every MFMA triple is pinned in program order and followed by one
// slice of the epilogue, so the matrix pipe works while the wave issues the slice.  Fill and drain bodies run
// the same code on clamped tiles with their side effects masked.
// Workgroup = 4 waves (128 edges per body), persistent over tile groups blockIdx.x, + gridDim.x, ...
// Weights stream through LDS in 32 KiB stages (16 per body: 8 of W_o, 8 of C), double buffered, loads two
// stages ahead, one barrier per stage.  The lo plane of the activation is parked in LDS (64 VGPRs less).
#include "edge_layer_common.h"

namespace difusco {

namespace pipe {
constexpr int H = 256;
constexpr int THREADS = 256;
constexpr int ENT = 512;                 // 32-byte rows per plane per stage
constexpr int PLANE = ENT * 16;          // 16-bit elements per plane per stage
constexpr int BUF = 2 * PLANE;           // one stage buffer: 2 planes = 32 KiB
constexpr int SCR_STRIDE = 36;                    // floats per edge row of the transposition scratch (32 + 4 pad)
constexpr int LDS_P = 7 * H * 4;                  // parameters                              7168 B at offset 0
constexpr int LDS_S = 4 * 32 * SCR_STRIDE * 4;    // per-wave scratch, 32 features per round 18432 B
constexpr int LDS_L = 4 * 16 * 1024;              // per-wave parked lo planes [slab][lane]  65536 B
constexpr int LDS_W = 2 * BUF * 2;                // two weight stage buffers                65536 B
constexpr int OFF_P = 0, OFF_S = LDS_P, OFF_L = OFF_S + LDS_S, OFF_W = OFF_L + LDS_L;
constexpr int LDS_TOTAL = OFF_W + LDS_W;          // 156672 of 163840
enum { P_BC = 0, P_GE, P_BE, P_T, P_GO, P_BO, P_BOUT };
}  // namespace pipe

template <typename T, bool STAMP, int ABL>
__global__ __launch_bounds__(256, 1) void edge_layer_pipe_kernel(
    float* e, const float* __restrict__ node4, const int* __restrict__ row, const int* __restrict__ col, int n_edges,
    int n_groups,   // 128-edge tile groups = ceil(n_edges / 128)
    const unsigned short* __restrict__ c_planes, const unsigned short* __restrict__ o_planes, long long plane_stride,
    const float* __restrict__ b_c, const float* __restrict__ g_e, const float* __restrict__ b_e,
    const float* __restrict__ tbias, const float* __restrict__ g_o, const float* __restrict__ b_o,
    const float* __restrict__ b_out, int time_on_edge, float* __restrict__ part, float* __restrict__ direct,
    unsigned long long* dbg) {   // dbg: phase timestamps of body 2 (STAMP only, profiling)
  using namespace pipe;
  typedef typename T::frag frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* prm = reinterpret_cast<float*>(smem_raw + OFF_P);

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned loff = lane * 4;
  // LDS addressing: every access is smem_raw + (one opaque per-lane byte offset) + (compile-time constant < 64 KiB),
  // i.e. one VGPR + the 16-bit immediate of the ds instruction.  Left to itself the compiler hoists hundreds of
  // loop-invariant "base + constant" address VGPRs out of the persistent loop and spills them.
  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  const int ob_prm = opaque(OFF_P + hh * 16);                                                  // + (param * H + fb0) * 4
  const int ob_scrw = opaque(OFF_S + wave * (32 * SCR_STRIDE * 4) + l31 * (SCR_STRIDE * 4) + hh * 16);   // m rows (write)
  const int ob_scrr = opaque(OFF_S + wave * (32 * SCR_STRIDE * 4) + l31 * 4);                  // column reads (lanes 0..31)
  const int ob_lo = opaque(OFF_L + wave * (16 * 1024) + lane * 16);                            // parked lo planes: + slab * 1024
#define PRM4(param, fb0) (*reinterpret_cast<const v4f*>(smem_raw + ob_prm + ((param) * H + (fb0)) * 4))
  const int n_it = (n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tile groups of this workgroup

  // layer parameters -> LDS (thread = feature)
  prm[P_BC * H + tid] = b_c[tid];
  prm[P_GE * H + tid] = g_e[tid];
  prm[P_BE * H + tid] = b_e[tid];
  prm[P_T * H + tid] = time_on_edge ? tbias[tid] : 0.0f;
  prm[P_GO * H + tid] = g_o[tid];
  prm[P_BO * H + tid] = b_o[tid];
  prm[P_BOUT * H + tid] = b_out[tid];

  // ---- weight stage stream: 16 stages per body, s < 8: W_o (output quarter s >> 1, k half s & 1), s >= 8: C
  //      (slabs 2 (s-8), 2 (s-8) + 1).  Chunk c of a stage: entry = c >> 1, half = c & 1; 4 chunks per thread.
  unsigned voff1[4], voff2[4];
  int ob_st[2][4];      // byte offset of chunk i in stage buffer 0 / 1
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + THREADS * i, entry = c >> 1, half = c & 1;
    voff1[i] = (entry >> 8) * 4096 + (entry & 255) * 16 + half * 8;
    voff2[i] = (entry >> 6) * 4096 + (entry & 63) * 16 + half * 8;
    ob_st[0][i] = opaque(OFF_W + wslot(entry, half) * 2);
    ob_st[1][i] = opaque(OFF_W + BUF * 2 + wslot(entry, half) * 2);
  }
  v4u wr[2][4];   // [plane][chunk]
  // the weight stream is the same every body: the per-thread offsets voff1/voff2 are made opaque once per body,
  // which keeps the compiler from hoisting all 16 stages of loads out of the persistent loop (and spilling them).
  // (Laundering the base POINTERS instead loses the global address space: the loads become flat_load, which also
  // count on lgkmcnt and stall every LDS wait.)
#define PIPE_LOAD_STAGE(s)                                                                              \
  {                                                                                                     \
    const unsigned short* sb = (s) < 8 ? o_planes + (long long)(8 * ((s) & 1)) * 4096 + 64 * ((s) >> 1) * 16 \
                                       : c_planes + (long long)(2 * ((s) - 8)) * 4096;        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      const unsigned vo = (s) < 8 ? voff2[i] : voff1[i];                                                \
      wr[0][i] = *reinterpret_cast<const v4u*>(sb + vo);                                                \
      wr[1][i] = *reinterpret_cast<const v4u*>(sb + plane_stride + vo);                                 \
    }                                                                                                   \
  }
#define PIPE_STORE_STAGE(s)                                                                             \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      *reinterpret_cast<v4u*>(smem_raw + ob_st[(s) & 1][i]) = wr[0][i];                                 \
      *reinterpret_cast<v4u*>(smem_raw + ob_st[(s) & 1][i] + PLANE * 2) = wr[1][i];                     \
    }                                                                                                   \
  }
  // stage s of the (periodic) stream parks stage s+1 in LDS and fetches stage s+2, in 8 pieces (chunk pc >> 1,
  // plane pc & 1) that ride between the MFMAs of the stage
#define PIPE_COPY(s, pc)                                                                                \
  if constexpr ((ABL & 32) == 0) {                                                                                                     \
    constexpr int s1_ = ((s) + 1) & 15, s2_ = ((s) + 2) & 15;                                           \
    const int i_ = (pc) >> 1, pl_ = (pc) & 1;                                                           \
    *reinterpret_cast<v4u*>(smem_raw + ob_st[s1_ & 1][i_] + pl_ * (PLANE * 2)) = wr[pl_][i_];           \
    const unsigned short* sb = s2_ < 8 ? o_planes + (long long)(8 * (s2_ & 1)) * 4096 + 64 * (s2_ >> 1) * 16 \
                                       : c_planes + (long long)(2 * (s2_ - 8)) * 4096;                  \
    const unsigned vo = s2_ < 8 ? voff2[i_] : voff1[i_];                                                \
    wr[pl_][i_] = *reinterpret_cast<const v4u*>(sb + pl_ * plane_stride + vo);                          \
  }
#define PIPE_END(s) __syncthreads();

  PIPE_LOAD_STAGE(0)
  PIPE_STORE_STAGE(0)
  PIPE_LOAD_STAGE(1)
  __syncthreads();

  int ob_wb[2];         // A-fragment read base of stage buffer 0 / 1 (bytes)
  ob_wb[0] = opaque(OFF_W + wslot(l31, hh) * 2);
  ob_wb[1] = opaque(OFF_W + BUF * 2 + wslot(l31, hh) * 2);
#define WFRAG(buf, elem_off) (*reinterpret_cast<const frag*>(smem_raw + ob_wb[buf] + (elem_off) * 2))
  const int last_tile = (n_edges - 1) >> 5;

  // ---- pipeline registers -----------------------------------------------------------------------------
  v16f accA[8];                  // GEMM 1 accumulators / epilogue registers of the current tile
  frag ph_[8][2];                // hi plane of the activation of the previous tile (B operand of GEMM 2); its lo plane is
                                 // parked in LDS (ob_lo): 64 VGPRs less in the stages where everything is live
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[nb][r] = 0.0f;
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      v4u z = {0u, 0u, 0u, 0u};
      ph_[nb][rg] = __builtin_bit_cast(frag, z);
      *reinterpret_cast<v4u*>(smem_raw + ob_lo + (2 * nb + rg) * 1024) = z;
    }
  }
  // edge context of the tile whose epilogue runs in the current body; fetched one body ahead
  int j_cur = 0, i_cur = 0;
  // neighbour-table rows (Ah[j], Bh[i], Vh[j]) of GA_RING quads in flight: quad qd lives in slot qd % GA_RING and
  // the rows of quad qd + GA_RING are requested right after quad qd is consumed (one stage = 4 quads ahead;
  // with a single wave per SIMD nothing else hides the ~2 us of a gather)
  constexpr int GA_RING = 4;
  v4f ga[GA_RING][3];
#define PIPE_GATHER_Q(qd, NJ, NI)                                                      \
  if constexpr ((ABL & 1) == 0) {                                                      \
    const int fb_ = 8 * (qd) + 4 * hh;                                                 \
    ga[(qd) % GA_RING][0] = *reinterpret_cast<const v4f*>((NJ) + 2 * H + fb_);         \
    ga[(qd) % GA_RING][1] = *reinterpret_cast<const v4f*>((NI) + 3 * H + fb_);         \
    ga[(qd) % GA_RING][2] = *reinterpret_cast<const v4f*>((NJ) + H + fb_);             \
  }
  {
    const float* n0 = node4;
#pragma unroll
    for (int qd = 0; qd < GA_RING; ++qd) PIPE_GATHER_Q(qd, n0, n0)
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
      for (int qd = 0; qd < GA_RING; ++qd)
#pragma unroll
        for (int u = 0; u < 3; ++u) ga[qd][u] = v4f{0.25f, 0.5f, 0.75f, 1.0f};
    }
  }

  constexpr float inv_h = 1.0f / 256.0f;

#define PIPE_STAMP(k)                                                                                  \
  if constexpr (STAMP) {                                                                               \
    if (kk == 2 && lane == 0) dbg[((long long)blockIdx.x * 4 + wave) * 32 + (k)] = __builtin_amdgcn_s_memtime(); \
  }
  for (int kk = -1; kk <= n_it; ++kk) {
    const bool do_e = kk >= 0 && kk < n_it, do_g2 = kk >= 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(voff1[i]), "+v"(voff2[i]));
    // tiles (clamped into the valid range when the pipeline is filling / draining: same code, masked effects)
    auto tile_of = [&](int k) {
      int g = (int)blockIdx.x + k * (int)gridDim.x;
      g = g < n_groups ? (g >= 0 ? g : 0) : n_groups - 1;
      return g * 4 + wave;
    };
    const int tile1 = tile_of(kk + 1), tileE = tile_of(kk), tile2 = tile_of(kk - 1);
    const int tile1c = tile1 < last_tile ? tile1 : last_tile;
    const float* e1 = e + (long long)tile1 * (32 * H);          // GEMM 1 operand stream (pad tiles exist, hold zeros)
    float* e2 = e + (long long)tile2 * (32 * H);                // residual read / store of GEMM 2
    const bool valid2 = do_g2 && (tile2 * 32 + l31) < n_edges;
    const bool validE = do_e && (tileE * 32 + l31) < n_edges;
    const float* nj = node4 + (long long)j_cur * 4 * H;
    const float* ni = node4 + (long long)i_cur * 4 * H;
    // neighbour-sum segment structure of the epilogue tile
    const int i_prev = __shfl_up(i_cur, 1, 64);
    const unsigned bnd = (unsigned)__ballot(l31 > 0 && i_cur != i_prev);
    const int first_end = bnd ? __builtin_ctz(bnd) : 32;
    float* part0 = part + ((long long)tileE * 2 + 0) * H;
    float* part1 = part + ((long long)tileE * 2 + 1) * H;
    // context of the NEXT epilogue tile (= the tile whose GEMM 1 runs now)
    const int s1e = tile1c * 32 + l31;
    const int s1c = s1e < n_edges ? s1e : n_edges - 1;
    const int j_next = col[s1c], i_next = row[s1c];

    float aggv = 0.0f, aggr[8];
    v4f bcq = PRM4(P_BC, 0);
    float s1 = 0.0f, q1 = 0.0f, s2 = 0.0f, q2s = 0.0f, mean1 = 0.0f, rstd1 = 0.0f, mean2 = 0.0f, rstd2 = 0.0f;
    v4f er[4][2];
    v16f acc2[2];
    v4f ein[2][4];

    // one quad of the epilogue: block nb = qd >> 2, quad g = qd & 3.  With a single wave per SIMD every LDS read
    // that is consumed in the slice that issued it stalls the whole instruction stream for the LDS latency, so the
    // b_C values of a quad (bcq) are requested one slice ahead (PIPE_BC_NEXT).
#define PIPE_BC_NEXT(qd) bcq = PRM4(P_BC, 8 * ((qd) & 31));
#define PIPE_QUAD(qd)                                                                                    \
  {                                                                                                      \
    constexpr int nb_ = (qd) >> 2;                                                                       \
    constexpr int g = (qd) & 3;                                                                          \
    const v4f bc = bcq;                                                                                  \
    const v4f ah = ga[(qd) % GA_RING][0], bh = ga[(qd) % GA_RING][1], vh = ga[(qd) % GA_RING][2];        \
    v4f m;                                                                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                      \
      const float ce = accA[nb_][4 * g + q] + bc[q];                                                     \
      const float ev = (ah[q] + bh[q]) + ce;                                                             \
      accA[nb_][4 * g + q] = ev;                                                                         \
      s1 += ev;                                                                                          \
      if constexpr ((ABL & 4) == 0) m[q] = validE ? fast_sigmoid(ev) * vh[q] : 0.0f; else m[q] = ev;      \
    }                                                                                                    \
    *reinterpret_cast<v4f*>(smem_raw + ob_scrw + (8 * g) * 4) = m;                                       \
  }
    // segmented column sums of the 32 features of block rnd (lanes 0..31 = features; see edge_layer.hip), in four
    // chunks of 8 edges; the 8 values of a chunk are requested one slice before they are summed (aggr[]), the
    // running sum aggv is carried from chunk to chunk
#define PIPE_AGG_LOAD(k0)                                                                                \
  if constexpr ((ABL & 2) == 0) {                                                                                                      \
    if ((k0) == 0) __builtin_amdgcn_wave_barrier();                                                      \
    _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                        \
      aggr[k] = *reinterpret_cast<const float*>(smem_raw + ob_scrr + ((k0) + k) * (SCR_STRIDE * 4));     \
  }
#define PIPE_AGG_SUM(rnd, k0)                                                                            \
  if constexpr ((ABL & 2) == 0) {                                                                                                      \
    if ((k0) == 0) aggv = 0.0f;                                                                          \
    const int f = 32 * (rnd) + l31;                                                                      \
    const bool wr_ok = do_e && hh == 0;                                                                  \
    _Pragma("unroll") for (int k2 = 0; k2 < 8; ++k2) {                                                   \
      const int k = (k0) + k2;                                                                           \
      if (k > 0 && ((bnd >> k) & 1u)) {                                                                  \
        const int node = __builtin_amdgcn_readlane(i_cur, k - 1);                                        \
        float* dst = (k == first_end) ? part0 : direct + (long long)node * H;                            \
        if (wr_ok) dst[f] = aggv;                                                                        \
        aggv = 0.0f;                                                                                     \
      }                                                                                                  \
      aggv += aggr[k2];                                                                                  \
    }                                                                                                    \
    if ((k0) == 24) {                                                                                    \
      float* dst = (first_end == 32) ? part0 : part1;                                                    \
      if (wr_ok) dst[f] = aggv;                                                                          \
      __builtin_amdgcn_wave_barrier();                                                                   \
    }                                                                                                    \
  }
    // e <- e + W_o a + b_o for output quarter qt (after its second stage)
#define PIPE_OUT(qt)                                                                                     \
  if ((ABL & 8) == 0 && valid2) {                                                                                          \
    _Pragma("unroll") for (int nbp = 0; nbp < 2; ++nbp)                                                  \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
        const v4f bo = PRM4(P_BOUT, 64 * (qt) + 32 * nbp + 8 * g);                                       \
        v4f v;                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) v[q] = ein[nbp][g][q] + (acc2[nbp][4 * g + q] + bo[q]); \
        *reinterpret_cast<v4f*>(e2 + ((4 * (qt) + 2 * nbp + (g >> 1)) * 512 + (g & 1) * 256) + loff) = v; \
      }                                                                                                  \
  }
    // LayerNorm / activation pieces of the epilogue, one accumulator block (16 values per lane) at a time
#define PIPE_CENTER(nb, MEAN, QACC)                                                    \
  {                                                                                    \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                   \
      const float d = accA[nb][r] - (MEAN);                                            \
      accA[nb][r] = d;                                                                 \
      QACC += d * d;                                                                   \
    }                                                                                  \
  }
#define PIPE_APPLY1(nb)                                                                \
  {                                                                                    \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                    \
      const v4f ge = PRM4(P_GE, 32 * (nb) + 8 * g);                                    \
      const v4f be = PRM4(P_BE, 32 * (nb) + 8 * g);                                    \
      const v4f tb = PRM4(P_T, 32 * (nb) + 8 * g);                                     \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                  \
        float y = accA[nb][4 * g + q] * rstd1 * ge[q] + be[q];                         \
        y = (y > 0.0f ? y : 0.0f) + tb[q];                                             \
        accA[nb][4 * g + q] = y;                                                       \
        s2 += y;                                                                       \
      }                                                                                \
    }                                                                                  \
  }
#define PIPE_CONVERT(nb, rg)                                                           \
  {                                                                                    \
    float a8[8];                                                                       \
    _Pragma("unroll") for (int g2 = 0; g2 < 2; ++g2) {                                 \
      const int g = 2 * (rg) + g2;                                                     \
      const v4f go = PRM4(P_GO, 32 * (nb) + 8 * g);                                    \
      const v4f bo = PRM4(P_BO, 32 * (nb) + 8 * g);                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                  \
        const float z = accA[nb][4 * g + q] * rstd2 * go[q] + bo[q];                   \
        a8[4 * g2 + q] = z * fast_sigmoid(z);                                          \
      }                                                                                \
    }                                                                                  \
    frag lo_;                                                                          \
    split8<T>(a8, ph_[nb][rg], lo_);                                                   \
    *reinterpret_cast<frag*>(smem_raw + ob_lo + (2 * (nb) + (rg)) * 1024) = lo_;       \
  }
    // what rides behind MFMA triple bi (0..15) of GEMM 2 stage s.  Stage s handles block nb = s of the epilogue
    // tile: its four quads (each consumed slot of the gather ring is refilled with the request for quad +
    // GA_RING), then the neighbour sum of the block in four chunks; the eight weight-copy pieces in between.
#define PIPE_REFILL(qd)                                     \
  if ((qd) + GA_RING < 32) PIPE_GATHER_Q((qd) + GA_RING, nj, ni)
#define PIPE_SLICE(s, bi)                                   \
  {                                                         \
    if ((bi) == 0) PIPE_QUAD(4 * (s) + 0)                   \
    if ((bi) == 1) PIPE_REFILL(4 * (s) + 0)                 \
    if ((bi) == 1) PIPE_BC_NEXT(4 * (s) + 1)                \
    if ((bi) == 2) PIPE_QUAD(4 * (s) + 1)                   \
    if ((bi) == 3) PIPE_REFILL(4 * (s) + 1)                 \
    if ((bi) == 3) PIPE_BC_NEXT(4 * (s) + 2)                \
    if ((bi) == 4) PIPE_QUAD(4 * (s) + 2)                   \
    if ((bi) == 5) PIPE_REFILL(4 * (s) + 2)                 \
    if ((bi) == 5) PIPE_BC_NEXT(4 * (s) + 3)                \
    if ((bi) == 6) PIPE_QUAD(4 * (s) + 3)                   \
    if ((bi) == 7) PIPE_REFILL(4 * (s) + 3)                 \
    if ((bi) == 7) PIPE_BC_NEXT(4 * (s) + 4)                \
    if ((bi) == 7) PIPE_AGG_LOAD(0)                         \
    if ((bi) == 8) PIPE_AGG_SUM(s, 0)                       \
    if ((bi) == 9) PIPE_AGG_LOAD(8)                         \
    if ((bi) == 10) PIPE_AGG_SUM(s, 8)                      \
    if ((bi) == 11) PIPE_AGG_LOAD(16)                       \
    if ((bi) == 12) PIPE_AGG_SUM(s, 16)                     \
    if ((bi) == 13) PIPE_AGG_LOAD(24)                       \
    if ((bi) == 14) PIPE_AGG_SUM(s, 24)                     \
    if (((bi) & 1) == 1) PIPE_COPY(s, (bi) >> 1)            \
  }

    // GEMM 2 stage s (0..7): output quarter qt = s >> 1, k half kc = s & 1, with the epilogue slices of stage s.
    // A single wave owns the SIMD, so consecutive MFMAs must not depend on each other: the two 32-feature output
    // blocks of the quarter (independent accumulators) are interleaved; each accumulator still sees its products in
    // the order lo*hi, hi*lo, hi*hi.
#define PIPE_G2(s)                                                                                       \
  {                                                                                                      \
    constexpr int qt = (s) >> 1, kc = (s) & 1;                                                           \
    if (kc == 0) {                                                                                       \
      _Pragma("unroll") for (int nbp = 0; nbp < 2; ++nbp)                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc2[nbp][r] = 0.0f;                              \
    } else if constexpr ((ABL & 8) == 0) {                                                               \
      _Pragma("unroll") for (int nbp = 0; nbp < 2; ++nbp)                                                \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                    \
          ein[nbp][g] = *reinterpret_cast<const v4f*>(e2 + ((4 * qt + 2 * nbp + (g >> 1)) * 512 + (g & 1) * 256) + loff); \
    }                                                                                                    \
    frag fh[4], fl[4], xlo[2];                                                                           \
    fh[0] = WFRAG((s) & 1, 0);                                                                           \
    fl[0] = WFRAG((s) & 1, PLANE);                                                                       \
    fh[1] = WFRAG((s) & 1, 32 * 16);                                                                     \
    fl[1] = WFRAG((s) & 1, PLANE + 32 * 16);                                                             \
    xlo[0] = *reinterpret_cast<const frag*>(smem_raw + ob_lo + (8 * kc) * 1024);                         \
    _Pragma("unroll") for (int pr = 0; pr < 8; ++pr) {   /* pr = k slab of the stage */                  \
      if (pr + 1 < 8) {                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                  \
          fh[(2 * pr + 2 + u) & 3] = WFRAG((s) & 1, ((pr + 1) * 64 + u * 32) * 16);                      \
          fl[(2 * pr + 2 + u) & 3] = WFRAG((s) & 1, PLANE + ((pr + 1) * 64 + u * 32) * 16);              \
        }                                                                                                \
        xlo[(pr + 1) & 1] = *reinterpret_cast<const frag*>(smem_raw + ob_lo + (8 * kc + pr + 1) * 1024); \
      }                                                                                                  \
      const int sl = 8 * kc + pr, a0 = (2 * pr) & 3, a1 = (2 * pr + 1) & 3;                              \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if constexpr ((ABL & 64) == 0) {                                                                   \
        acc2[0] = T::mfma(fl[a0], ph_[sl >> 1][sl & 1], acc2[0]);                                        \
        acc2[1] = T::mfma(fl[a1], ph_[sl >> 1][sl & 1], acc2[1]);                                        \
        acc2[0] = T::mfma(fh[a0], xlo[pr & 1], acc2[0]);                                                 \
      }                                                                                                  \
      PIPE_SLICE(s, 2 * pr)                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      if constexpr ((ABL & 64) == 0) {                                                                   \
        acc2[1] = T::mfma(fh[a1], xlo[pr & 1], acc2[1]);                                                 \
        acc2[0] = T::mfma(fh[a0], ph_[sl >> 1][sl & 1], acc2[0]);                                        \
        acc2[1] = T::mfma(fh[a1], ph_[sl >> 1][sl & 1], acc2[1]);                                        \
      }                                                                                                  \
      PIPE_SLICE(s, 2 * pr + 1)                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
  }
    // GEMM 1 stage t (0..7): slabs 2t, 2t+1 of C against the e stream of the next tile; output blocks in
    // interleaved pairs as above
#define PIPE_G1(t)                                                                                       \
  {                                                                                                      \
    frag xh[2], xl[2];                                                                                   \
    _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                \
      const int ks = 2 * (t) + sub;                                                                      \
      const v4f c0 = er[ks % 4][0], c1 = er[ks % 4][1];                                                  \
      if ((ABL & 128) == 0 && ks + 4 < 16) {                                                             \
        er[ks % 4][0] = *reinterpret_cast<const v4f*>(e1 + (ks + 4) * 512 + loff);                       \
        er[ks % 4][1] = *reinterpret_cast<const v4f*>(e1 + ((ks + 4) * 512 + 256) + loff);               \
      }                                                                                                  \
      const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};                      \
      split8<T>(xs, xh[sub], xl[sub]);                                                                   \
    }                                                                                                    \
    frag fh[4], fl[4];                                                                                   \
    fh[0] = WFRAG((t) & 1, 0);                                                                           \
    fl[0] = WFRAG((t) & 1, PLANE);                                                                       \
    fh[1] = WFRAG((t) & 1, 32 * 16);                                                                     \
    fl[1] = WFRAG((t) & 1, PLANE + 32 * 16);                                                             \
    _Pragma("unroll") for (int pr = 0; pr < 8; ++pr) {   /* blocks 2 pr, 2 pr + 1 of the 16 of the stage */ \
      if (pr + 1 < 8) {                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                  \
          const int bn = 2 * pr + 2 + u;                                                                 \
          fh[bn & 3] = WFRAG((t) & 1, (bn >> 3) * 256 * 16 + (bn & 7) * 32 * 16);                        \
          fl[bn & 3] = WFRAG((t) & 1, PLANE + (bn >> 3) * 256 * 16 + (bn & 7) * 32 * 16);                \
        }                                                                                                \
      }                                                                                                  \
      const int b0 = 2 * pr, nb = b0 & 7, sub = b0 >> 3, a0 = b0 & 3, a1 = (b0 + 1) & 3;                 \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      accA[nb] = T::mfma(fl[a0], xh[sub], accA[nb]);                                                     \
      accA[nb + 1] = T::mfma(fl[a1], xh[sub], accA[nb + 1]);                                             \
      accA[nb] = T::mfma(fh[a0], xl[sub], accA[nb]);                                                     \
      accA[nb + 1] = T::mfma(fh[a1], xl[sub], accA[nb + 1]);                                             \
      accA[nb] = T::mfma(fh[a0], xh[sub], accA[nb]);                                                     \
      accA[nb + 1] = T::mfma(fh[a1], xh[sub], accA[nb + 1]);                                             \
      PIPE_COPY(8 + (t), pr)                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
  }

    // ======================= stages 0..7 : GEMM 2 (tile kk-1)  |  epilogue part 1 (tile kk) ================
#define PIPE_STAGE_A(s)                                           \
  PIPE_STAMP(s)                                                   \
  if ((s) == 2 || (s) == 3) PIPE_STAMP(18 + 5 * ((s) - 2))        \
  PIPE_G2(s)                                                      \
  if ((s) == 2 || (s) == 3) PIPE_STAMP(19 + 5 * ((s) - 2))        \
  if ((s) == 2 || (s) == 3) PIPE_STAMP(20 + 5 * ((s) - 2))        \
  if (((s) & 1) == 1) {                                           \
    PIPE_OUT((s) >> 1)                                            \
  }                                                               \
  if ((s) == 2 || (s) == 3) PIPE_STAMP(21 + 5 * ((s) - 2))        \
  PIPE_END(s)
    PIPE_STAGE_A(0)
    PIPE_STAGE_A(1)
    PIPE_STAGE_A(2)
    PIPE_STAGE_A(3)
    PIPE_STAGE_A(4)
    PIPE_STAGE_A(5)
    PIPE_STAGE_A(6)
    PIPE_STAGE_A(7)
#undef PIPE_STAGE_A

    PIPE_STAMP(8)
    // the e stream of the next tile starts now: four slabs ahead of GEMM 1
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if constexpr ((ABL & 128) == 0) {
        er[d][0] = *reinterpret_cast<const v4f*>(e1 + d * 512 + loff);
        er[d][1] = *reinterpret_cast<const v4f*>(e1 + (d * 512 + 256) + loff);
      } else {
        er[d][0] = v4f{0.1f, 0.2f, 0.3f, 0.4f} * (float)lane;
        er[d][1] = v4f{0.5f, 0.6f, 0.7f, 0.8f} * (float)lane;
      }
    }

    // ======================= epilogue part 2 (tile kk): LayerNorm_e, ReLU, + t, LayerNorm_o, SiLU, split =====
    if constexpr ((ABL & 16) == 0) {
      mean1 = (s1 + __shfl_xor(s1, 32, 64)) * inv_h;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) PIPE_CENTER(nb, mean1, q1)
      rstd1 = __builtin_amdgcn_rsqf((q1 + __shfl_xor(q1, 32, 64)) * inv_h + 1e-5f);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) PIPE_APPLY1(nb)
      mean2 = (s2 + __shfl_xor(s2, 32, 64)) * inv_h;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) PIPE_CENTER(nb, mean2, q2s)
      rstd2 = __builtin_amdgcn_rsqf((q2s + __shfl_xor(q2s, 32, 64)) * inv_h + 1e-5f);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        PIPE_CONVERT(nb, 0)
        PIPE_CONVERT(nb, 1)
      }
    } else {   // profiling only: planes straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
          float a8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) a8[q] = accA[nb][8 * rg + q] + s1;
          frag lo_;
          split8<T>(a8, ph_[nb][rg], lo_);
          *reinterpret_cast<frag*>(smem_raw + ob_lo + (2 * nb + rg) * 1024) = lo_;
        }
    }

    // ======================= stages 8..15 : GEMM 1 (tile kk+1) ==============================================
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[nb][r] = 0.0f;
#define PIPE_STAGE_B(t)  \
  PIPE_STAMP(9 + (t))    \
  PIPE_G1(t)             \
  PIPE_END(8 + (t))
    PIPE_STAGE_B(0)
    PIPE_STAGE_B(1)
    PIPE_STAGE_B(2)
    PIPE_STAGE_B(3)
    PIPE_STAGE_B(4)
    PIPE_STAGE_B(5)
    PIPE_STAGE_B(6)
    {   // first quads of the next epilogue tile: requested under the last GEMM 1 stage
      const float* njn = node4 + (long long)j_next * 4 * H;
      const float* nin = node4 + (long long)i_next * 4 * H;
#pragma unroll
      for (int qd = 0; qd < GA_RING; ++qd) PIPE_GATHER_Q(qd, njn, nin)
    }
    PIPE_STAGE_B(7)
#undef PIPE_STAGE_B

    PIPE_STAMP(17)
    // the tile whose GEMM 1 just ran enters its epilogue in the next body
    j_cur = j_next;
    i_cur = i_next;
  }
#undef PRM4
#undef WFRAG
#undef PIPE_GATHER_Q
#undef PIPE_QUAD
#undef PIPE_AGG_LOAD
#undef PIPE_AGG_SUM
#undef PIPE_BC_NEXT
#undef PIPE_COPY
#undef PIPE_G2
#undef PIPE_OUT
#undef PIPE_G1
#undef PIPE_CENTER
#undef PIPE_APPLY1
#undef PIPE_CONVERT
#undef PIPE_SLICE
#undef PIPE_STAMP
#undef PIPE_REFILL
#undef PIPE_LOAD_STAGE
#undef PIPE_STORE_STAGE
#undef PIPE_END
}

int g_fused_variant = 0;   // 0: edge_layer_fused_kernel (phase-serial), 1: edge_layer_pipe_kernel (software pipelined)

template <typename T, bool STAMP, int ABL = 0>
static hipError_t launch_pipe_t(float* e, const float* node4, const int* row, const int* col, int n_edges,
                                const unsigned short* c_planes, const unsigned short* o_planes, long long plane_stride,
                                const float* b_c, const float* g_e, const float* b_e, const float* tbias,
                                const float* g_o, const float* b_o, const float* b_out, int time_on_edge, float* part,
                                float* direct, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_layer_pipe_kernel<T, STAMP, ABL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, pipe::LDS_TOTAL);
    if (er != hipSuccess) return er;
    attr_set = true;
  }
  const int n_groups = (n_edges + 127) / 128;
  const int grid = n_groups < 256 ? n_groups : 256;          // one persistent workgroup per CU
  hipLaunchKernelGGL((edge_layer_pipe_kernel<T, STAMP, ABL>), dim3(grid), dim3(256), pipe::LDS_TOTAL, stream, e, node4, row, col,
                     n_edges, n_groups, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o, b_o, b_out,
                     time_on_edge, part, direct, g_fused_dbg);
  return hipGetLastError();
}

hipError_t launch_edge_layer_pipe(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                  const unsigned short* c_planes, const unsigned short* o_planes, long long plane_stride,
                                  const float* b_c, const float* g_e, const float* b_e, const float* tbias,
                                  const float* g_o, const float* b_o, const float* b_out, int time_on_edge, float* part,
                                  float* direct, hipStream_t stream) {
  if (n_edges <= 0) return hipSuccess;
#define PIPE_ARGS e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o, b_o, b_out, \
                  time_on_edge, part, direct, stream
  if (mode == 3 && g_fused_dbg) return launch_pipe_t<FFp16, true>(PIPE_ARGS);
  if (mode == 3 && g_fused_ablate != 0) {   // profiling-only variants (results are wrong by construction)
    switch (g_fused_ablate) {
      case 1: return launch_pipe_t<FFp16, false, 1>(PIPE_ARGS);
      case 2: return launch_pipe_t<FFp16, false, 2>(PIPE_ARGS);
      case 4: return launch_pipe_t<FFp16, false, 4>(PIPE_ARGS);
      case 8: return launch_pipe_t<FFp16, false, 8>(PIPE_ARGS);
      case 16: return launch_pipe_t<FFp16, false, 16>(PIPE_ARGS);
      case 32: return launch_pipe_t<FFp16, false, 32>(PIPE_ARGS);
      case 64: return launch_pipe_t<FFp16, false, 64>(PIPE_ARGS);
      case 128: return launch_pipe_t<FFp16, false, 128>(PIPE_ARGS);
      case 7: return launch_pipe_t<FFp16, false, 7>(PIPE_ARGS);
      default: return hipErrorInvalidValue;
    }
  }
  if (mode == 1)
    return launch_pipe_t<FBf16, false>(e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o,
                                b_o, b_out, time_on_edge, part, direct, stream);
  if (mode == 3)
    return launch_pipe_t<FFp16, false>(e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o,
                                b_o, b_out, time_on_edge, part, direct, stream);
  return hipErrorInvalidValue;
}

}  // namespace difusco
