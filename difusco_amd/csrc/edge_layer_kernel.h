// Fused edge pass of one DIFUSCO GNN layer for gfx950 (H = 256): the edge state e is read ONCE and
// written ONCE per layer.
//
//   Ce    = C e                                    gnn_encoder.py:104           GEMM 1 (matrix cores)
//   e'    = Ah[j] + Bh[i] + (Ce + b_C)             :110      (carried as e' log2(e): the node rows arrive pre-multiplied with b_C
//                                                             folded in, see "LOG2E DOMAIN" at the kernel; difusco_hip.h, ABI 11)
//   m     = sigmoid(e') * Vh[j]  -> sum over the edges of centre node i          :112,:115,:163,:177-191
//   y     = ReLU(LN_e(e')) (+ t_l, TSP)            :131,:135,:445
//   a     = SiLU(LN_o(y))                          per_layer_out[l][0:2] :339-342
//   e    += W_o a + b_o                            per_layer_out[l][2] :344, residual :449   GEMM 2
//
// Chained, transposed MFMA.  Both GEMMs are computed as D[f][edge] = sum_k W[f][k] X[edge][k] (A operand =
// 32 weight rows, B operand = 32 edges), so in the 32x32 accumulator a lane owns, for ITS edge (lane&31),
// the features {32 nb + 8 g + 4 hh + 0..3} - half of the 256 features, the other half sits in lane+32.
//   * LayerNorm over the 256 features of an edge = in-lane sum over 128 registers + ONE cross-half exchange.
//   * Eight consecutive accumulator registers of a block are exactly the B operand (8 k values per lane
//     half) of a 32x32x16 MFMA: the epilogued accumulators of GEMM 1 feed GEMM 2 straight from registers,
//     no LDS round trip.  The weight planes are stored in that k order (weights.py: slab position order
//     {0..3, 8..11, 4..7, 12..15}).
//   * fp32 operands are split into two 16-bit planes (fp16 planes of power-of-two-scaled operands - see `scales` at the
//     kernel - or bf16) and multiplied with 3 MFMA products (hi*hi, hi*lo, lo*hi), accumulated in fp32 (see linear_split.hip).
// A wave owns a 32-edge tile end to end; a workgroup is 4 waves (two workgroups per CU, whose phases drift
// apart so that one's MFMA phases overlap the other's VALU / address-heavy epilogue) - see fused::Geo.
// Weights stream through LDS in stages of 256 rows of 32 bytes x 2 planes ([256 rows][16 k] slabs for GEMM 1,
// [64 rows][16 k] sub-slabs of one output quarter for GEMM 2), double buffered, filled by LDS-DMA
// (buffer_load_dwordx4 ... lds, no staging registers), one barrier per stage; the rows are XOR-swizzled so every
// 16-lane group of ds_read_b128 hits 16 distinct 16-byte bank slots without padding (the swizzle is applied to the
// per-lane source address of the DMA, whose LDS side is lane-linear).  GEMM 2 runs in four quarters of 64 output features (32
// accumulator registers) so that act planes (128) + accumulators + staging fit 256 VGPRs (2 waves/SIMD).
//
// Neighbour-table rows A h[j], V h[j] reach the lanes by FULL-LINE gathers: LDS-DMA into the wave's idle share of a weight stage
// buffer, conflict-free ds_read_b128 back (OPT bit 14 at the gather phase); B h[i] by plain buffer loads.
//
// Neighbour sum.  The gated messages m of a tile are transposed through a wave-private LDS scratch
// (64 features per round) and summed per centre-node segment by lanes = features.  A segment that is
// the first or last of its tile may continue in the neighbouring tile: it goes to part[tile][0|1];
// segments strictly inside a tile are complete and go to direct[node].  node_finalize_kernel adds the
// pieces of each node in tile order - deterministic, no atomics.
//
// First layer (template flag L0): when the edge input is a lookup in a 2-row table (categorical TSP: the embedding
// of the bit x_t; MIS: zeros) the table sits in LDS and the kernel never reads e - see the L0 notes at the kernel.
// This header holds the kernel template and its launcher template; the instantiations are spread over
// edge_layer.hip (fp16 production variants), edge_layer_bf16.hip and edge_layer_abl.hip (profiling-only ablations) so
// that the translation units compile in parallel.
#pragma once
#include "edge_layer_common.h"

namespace difusco {

namespace fused {
constexpr int H = 256;
constexpr int SCR_STRIDE = 68;           // floats per edge row of the aggregation scratch (64 + 4 pad)
enum { P_GE = 0, P_BE, P_T, P_GO, P_BO, P_BOUT, P_TAB0, P_TAB1, P_CE0, P_CE1, P_COUNT };
// P_TAB*: layer-0 input table rows; P_CE*: C (weight of GEMM 1) applied to those rows

// Workgroup geometry (template parameter NW = 4, the only one kept): a workgroup is 4 tiles of 32 edges, a weight stage
// holds ENT = 256 rows of 32 bytes per plane (16 KiB for the two planes, 32 stages per tile), TWO workgroups per CU
// whose phases drift apart, so that one's MFMA phases overlap the other's VALU / address-unit heavy epilogue.  Measured
// and dropped (profiles/r01, r02 fused_kernel_study.txt): one 8-wave workgroup per CU (half the L2 weight traffic, phases
// in lock step: 8 % slower) and 32 KiB stages with the scratch aliasing a stage buffer (half the barriers: 2-4 % slower).
template <int CODE>
struct Geo {
  static_assert(CODE == 4, "one workgroup geometry");
  static constexpr int WAVES = 4;
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int ENT = 256;               // entries (32-byte rows) per plane per stage
  static constexpr int PP = ENT / 32 / WAVES;   // 1 KiB LDS-DMA pieces per wave, plane and stage (2)
  static constexpr int PLANE = ENT * 16;        // 16-bit elements per plane per stage
  static constexpr int BUF = 2 * PLANE;         // one stage buffer: 2 planes
  static constexpr int SPS = ENT / 256;         // GEMM 1: slabs per stage (1)
  static constexpr int NS1 = 16 / SPS;          // GEMM 1 stages (16)
  static constexpr int KPS = ENT / 64;          // GEMM 2: k slabs per stage (4)
  static constexpr int SPQ = 16 / KPS;          // GEMM 2: stages per output quarter (4)
  static constexpr int NSTAGE = NS1 + 4 * SPQ;  // 32
  static constexpr int LDS_W = 2 * BUF * 2;     // bytes, double buffered                               32768
  static constexpr int LDS_P = P_COUNT * H * 4; // bytes: g_e, b_e, t, g_o, b_o, b_O, table rows (4)      10240
  static constexpr int LDS_S = WAVES * 32 * SCR_STRIDE * 4;   // bytes                                  34816
  static constexpr int OFF_P = LDS_W;           // [buffers][parameters][scratch]
  static constexpr int OFF_S = LDS_W + LDS_P;
  static constexpr int LDS_TOTAL = LDS_W + LDS_P + LDS_S;     //                                        77824
};
constexpr int geo_waves(int) { return 4; }
}  // namespace fused

template <typename T, int ABL, int NW, bool L0, bool GNP, int TAIL, int OPT, bool NOTB = false>
__global__ __launch_bounds__(64 * fused::geo_waves(NW), 2) void edge_layer_fused_kernel(
    float* e, const float* __restrict__ node4, const int* __restrict__ row, const int* __restrict__ col, int n_edges,
    const unsigned short* __restrict__ c_planes, const unsigned short* __restrict__ o_planes, long long plane_stride,
    const float* __restrict__ b_c, const float* __restrict__ g_e, const float* __restrict__ b_e,
    const float* __restrict__ tbias, const float* __restrict__ g_o, const float* __restrict__ b_o,
    const float* __restrict__ b_out, int time_on_edge, float* __restrict__ part, float* __restrict__ direct,
    unsigned long long* dbg,     // dbg: optional phase timestamps (profiling), nullptr in production
    const float* __restrict__ l0_table, const float* __restrict__ l0_x, const int* __restrict__ l0_perm,
    float* __restrict__ gn_tile, const float* __restrict__ scales, const float* __restrict__ etmax_in,
    float* __restrict__ etmax_out, int start_delay) {
  // Operand scaling (T::kScaled, i.e. fp16 planes; see edge_layer_common.h).  scales = {2^-kc, 2^-(ko+ka), 2^ka, -log2(e) 2^-ka}:
  // the weight planes of C / W_o hold W 2^kc / W 2^ko (weights.py), the activation of GEMM 2 is produced as a 2^ka with ka
  // from the host-side bound |a| <= 16 max|g_o| + max|b_o| (LayerNorm output is bounded by sqrt(H - 1)), and the e stream
  // of GEMM 1 is scaled per 32-edge tile by 2^kx, kx from the tile's max|e| that the PRODUCER of e left in etmax_in[tile]
  // (the previous layer's output phase writes etmax_out, the embedding kernels do the same).  Every scale is a power of
  // two, so products and sums are the exact scaled values and one multiply in the epilogue (fused into the bias add)
  // restores them: the two-plane split then keeps its 22 significand bits for any finite fp32 operand scale.
  // TAIL: what the step still needs from this layer.  0 = everything.  1 = last layer of a TSP step: the head reads
  // only e, so the node update is dead work - no V h gathers, no gate, no neighbour sum (the caller skips
  // node_finalize).  2 = last layer of a MIS step: the head reads only h, so the edge output is dead work - the kernel
  // ends after the neighbour sum (no LayerNorms, no GEMM 2, no store of e).  The reference computes both and discards
  // them (gnn_encoder.py:400-401 / :412-413 read one of the two states).
  // GNP (last layer of a step whose head reads e): per tile and GroupNorm group (8 channels = the two lane halves of
  // one (quarter, block, quad)), the sum and the sum of squares of the NEW e values go to gn_tile[tile][32][2]; the
  // head then needs no statistics pass over e (nn.py:93-100, gnn_encoder.py:400-401).
  // L0 (first layer of a step whose edge input is a table lookup): e_in[s] = l0_table[x > 0.5 ? 1 : 0] with
  // x = l0_x[l0_perm ? l0_perm[s] : s] (categorical TSP: the edge embedding of the bit x_t, gnn_encoder.py:395) or
  // row 0 when l0_x is null (MIS: e = zeros, gnn_encoder.py:407).  The kernel then never reads e, and it has no GEMM 1
  // either: with only two distinct input rows, C e_in is one of two vectors (l0_table rows 2, 3 = C applied to rows 0,
  // 1, computed once per step by an exact fp32 linear on two rows); the accumulators start from that row, the
  // residual comes from the input row, and the separate embedding pass over e disappears.
  // OPT: scheduling / code-shape options, exact except bits 6 and 11 (see the end of this list); A/B through
  // difusco_debug_set(7, ..):
  //   bit 0  XCD-contiguous tile ranges: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only); the
  //          remap gives every XCD one contiguous range of tiles, so that - with the nodes of a graph in Morton order
  //          (graph.py) - the neighbour-table rows A h[j], V h[j] gathered by one XCD's workgroups are a small, spatially
  //          compact set that stays in that XCD's 4 MiB L2;
  //   bit 1  the MFMAs of two weight blocks are issued alternately (two independent accumulator chains), so that no
  //          MFMA waits for the result of the one issued right before it.  Part of the production set in rounds 2-4; OFF since
  //          round 5: in the power-limited regime the three products of a block back to back measure 0.3-0.5 % faster on every
  //          workload (profiles/r05/ab_mfma_chain_alternation_same_box.txt); per accumulator the order is the same: bit-identical.
  //   bit 4  non-temporal accesses for the e stream (see kNt below).
  //   bit 5  two stages of cover for the e stream (kDeepE below);  bit 6  LayerNorm reductions as four partial sums (required).
  //   bit 8  GEMM 1 input slabs of e through a buffer resource (kBufRing below).
  //   bit 9  raised issue priority outside the GEMM phases;  bit 10  default cache policy for the GEMM 1 slabs of e.
  //   bit 11 element-wise arithmetic on register pairs (required: the scalar form was removed in round 5).
  //   bit 12 persistent workgroups (measured slower: -2.0 % round 3, -4.7 % round 5; profiling library only);
  //   bit 14 FULL-LINE neighbour-table gathers through LDS-DMA (+3.4 %, see the gather phase), bit 15 two units + counted waits (no
  //          further gain, profiling library only).
  //   bit 17 neighbour-sum fast path for tiles with one centre node (round 4) and, on the other tiles, segment-start tests only inside
  //          the 8-row groups that hold one (round 5); both bit-identical.
  //   (removed in round 5, all measured slower - HISTORY.md: bit 13 gathers two batches ahead, bit 16 full-line gathers through
  //    staging registers, bit 18 16-bit planes of a product by v_fma_mix, and the forms without bits 6 / 11)
  // Production = 151409 (bits 0, 4, 5, 6, 8, 9, 10, 11, 14, 17).  Bits 0-11 = 3955: +16 % on the step over 0 (profiles/r02/fused_kernel_study.txt; that file also
  // records two restructurings that were measured and removed - next-stage requests issued between the MFMA groups,
  // and A h[j] + B h[i] gathered straight into the GEMM 1 accumulators).  Bits 0-5, 8-10 do not change a result bit; bit 6
  // changes the summation order of the LayerNorm statistics and bit 11 where the compiler contracts multiply-adds (fp32
  // rounding, ~1e-6 on e; 9 % of the elements move by one or two ulps between 1907 and 3955).
  // ABL: profiling-only ablation mask, 0 in production (bit0 no gathers, bit1 no neighbour sum,
  // bit2 no LN/activation math, bit3 no GEMM 2); compile-time so that it cannot perturb the real kernel
  constexpr int ablate = ABL;
  using namespace fused;
  typedef Geo<NW> G_;
  constexpr int WAVES = G_::WAVES, PLANE = G_::PLANE, BUF = G_::BUF, NSTAGE = G_::NSTAGE, SPS = G_::SPS, NS1 = G_::NS1,
                KPS = G_::KPS, SPQ = G_::SPQ, PP = G_::PP;
  typedef typename T::frag frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wbuf = reinterpret_cast<unsigned short*>(smem_raw);
  float* prm = reinterpret_cast<float*>(smem_raw + G_::OFF_P);
  float* scr_all = reinterpret_cast<float*>(smem_raw + G_::OFF_S);

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  if constexpr ((OPT & 1) != 0) {      // bijective for any grid size (cdna_hip_programming.md T1)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  // OPT bit 12: PERSISTENT workgroups.  The grid is 2 workgroups per CU (the residency the 78 KB of LDS allow); workgroup
  // (XCD x = blockIdx % 8, position idx) walks the 128-edge groups start_x + idx, + q, + 2q, ... of the contiguous range of
  // its XCD (the same locality as bit 0).  The layer parameters are written to LDS once, the weight-stage ring simply
  // continues into the next group (its first stage is requested during the last stage of the current one), and the first
  // e slabs / the tile scale of the next group are requested before the last output phase: the per-tile prologue (5.5 k
  // of 122 k cycles: first-stage and e latency, parameter fill) disappears.  A launch with no more groups than workgroups
  // runs every workgroup once, exactly like the non-persistent kernel.
  constexpr bool kPersist = (OPT & 4096) != 0;
  const int n_wgt = (n_edges + 32 * WAVES - 1) / (32 * WAVES);
  int wt = bid, wt_end = bid + 1, wt_step = 1;
  if (kPersist && (int)gridDim.x < n_wgt) {
    const int q = (int)gridDim.x >> 3, x = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;      // (grid: a multiple of 8)
    const int base = n_wgt >> 3, rem = n_wgt & 7;
    const int start = x * base + (x < rem ? x : rem);
    wt = start + idx;
    wt_end = start + base + (x < rem ? 1 : 0);
    wt_step = q;
  }
  // start_delay > 0 (experiment, profiling library): the workgroups that take the SECOND slot of every CU in the first
  // dispatch generation (positions 32..63 of their XCD) start that many cycles late, so that the two resident workgroups of
  // a CU begin out of phase (one in its matrix phases while the other is in its VALU phases); later generations inherit
  // the offset because a workgroup starts when its predecessor in the slot ends.
  if (start_delay > 0) {
    const int pos = (int)blockIdx.x >> 3;
    if (pos >= 32 && pos < 64) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)start_delay) __builtin_amdgcn_s_sleep(32);
    }
  }
  const unsigned loff = lane * 4;
  const int loff_b = lane * 16;
  float* scr = scr_all + wave * 32 * SCR_STRIDE;
  typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
  // OPT bit 4: the e stream (read once, written once per layer: 1.6 GB per launch) moves with non-temporal accesses, so
  // that it does not push the neighbour-table rows and the weight planes - both re-read by every workgroup of the
  // XCD - out of the 4 MiB L2
  constexpr bool kNt = (OPT & 16) != 0;
  // OPT bit 8: the GEMM 1 input slabs of e are read through a buffer resource over the wave's tile (wave-uniform base in
  // SGPRs, one 32-bit lane offset, the slab offset in an SGPR) instead of per-lane 64-bit addresses.  Measured together
  // with the iterative-maxocc instruction scheduler (build.py): no scratch in any variant and +3.7 % on the step.  The
  // residual loads and the stores keep the global (scalar base + 32-bit offset) form: routing them through the buffer
  // as well measured the same, and buffer_store_dwordx4 with an SGPR soffset needs a hand-placed wait state on gfx950
  // (profiles/r02/fused_kernel_study.txt, "buffer stores").
  constexpr bool kBufRing = (OPT & 256) != 0;
  // OPT bit 10: the GEMM 1 slabs are read with the default cache policy; the residual re-read (last use) and the stores stay
  // non-temporal.  Measured on one box, graph-steps/s: all three non-temporal 785.1 | slabs default 802.6 | residual default
  // 765.4 | slabs + residual default 782.5 | stores default 759.4 | slabs + stores default 774.9.
  constexpr bool kNtRing = kNt && (OPT & 1024) == 0, kNtRes = kNt, kNtSt = kNt;
  // profiling only (wrong results): matrix phases without stage refills and barriers / without the e stream
  constexpr bool kNoSync = (ABL & 16384) != 0, kNoE = (ABL & 32768) != 0;
  // profiling only (races, wrong results): stage requests never waited for / no stage barrier
  constexpr bool kNoWait = (ABL & 65536) != 0, kNoBar = (ABL & 131072) != 0;
  // B operand of GEMM 1: slab ks needs e[s][16 ks + {4hh..4hh+3, 8+4hh..8+4hh+3}] = two float4 of the tiled
  // layout = 2 KiB per wave and slab, cold HBM reads.
  // A register ring, RING slabs ahead of the MFMAs.  (Routing this stream through LDS-DMA into the idle aggregation
  // scratch was measured 2.5 % slower - profiles/r01/fused_kernel_study.txt - and removed.)
  constexpr int RING = 2;
  v4f er[RING][2];
  // the first RING slabs of tile `tl` (e is TILED, see below) into the ring
  auto ring_fill = [&](int tl) {
    float* const et = e + (long long)tl * (32 * H);
    if constexpr (kBufRing) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(et, 0, 32 * H * 4, 0x00020000);
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        er[d][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, loff_b, d * 2048, kNtRing ? 2 : 0));
        er[d][1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, loff_b, d * 2048 + 1024, kNtRing ? 2 : 0));
      }
    } else {
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        er[d][0] = *reinterpret_cast<const v4f*>(et + d * 512 + loff);
        er[d][1] = *reinterpret_cast<const v4f*>(et + d * 512 + 256 + loff);
      }
    }
  };
  float tmax_cur = 0.0f;      // max |e| of the current tile (kScaled; left by the producer of e in etmax_in)

  // ---- weight stage streaming ---------------------------------------------------------------------
  // A stage is ENT rows of 32 bytes per plane (16 KiB for the two planes):
  //   stage t < NS1 : GEMM 1, slab t of C;  entry = weight row
  //   stage NS1 + u : GEMM 2, output quarter qt = u / SPQ (64 features), slabs KPS kc .. KPS kc + KPS - 1 of W_o (kc = u % SPQ);
  //                   entry = ksl * 64 + (row - 64 qt)
  // Stage t lives in LDS buffer t & 1 and is filled by LDS-DMA: buffer_load_dwordx4 ... lds moves 1 KiB per wave
  // instruction from global memory straight into the stage buffer - no staging registers, no ds_write.  The LDS image
  // is swizzled (wslot): lane L of wave w, piece i fills slot (PP w + i) * 64 + L of a plane, i.e.
  // entry = (PP w + i) * 32 + (L >> 1), and fetches the half that belongs there (the XOR of wslot applied on the
  // source side; both halves of an entry are adjacent in global memory, so coalescing is unchanged).
  // Protocol: iteration t requests stage t+1 into the other buffer (everybody left it at the last barrier),
  // multiplies stage t, then waits for its own requests before the barrier.  The last stage of a tile (LAST_T) requests
  // the FIRST stage of the workgroup's next tile (persistent kernel; the stage count per tile is even, so the buffer
  // parity carries over) - that request is waited for at the top of the next tile.
  unsigned dvoff1, dvoff2;
  {
    const int entry0 = (PP * wave) * 32 + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
    dvoff1 = (entry0 >> 8) * 4096 + (entry0 & 255) * 16 + half * 8;
    dvoff2 = (entry0 >> 6) * 4096 + (entry0 & 63) * 16 + half * 8;
  }
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c_planes), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(o_planes), 0, 0x7fffffff, 0x00020000);
  const int plane_bytes = (int)plane_stride * 2;
  constexpr bool kLate16 = (OPT & 16384) != 0 && (OPT & 32768) != 0 && TAIL != 2;      // see OPT bit 15 at the gather phase
  constexpr int FIRST_T = L0 ? NS1 : 0;                       // first stage of a tile (layer 0 has no GEMM 1)
  constexpr int LAST_T = TAIL == 2 ? NS1 - 1 : NSTAGE - 1;    // last stage of a tile (MIS last layer: no GEMM 2)
  static_assert(((LAST_T + 1 - FIRST_T) & 1) == 0, "an even number of stages per tile keeps the buffer parity");
  // piece i of this wave (i < PP): LDS slot block PP*wave + i; its source lies i * 512 elements further in a GEMM 1
  // stage ([slab][256 rows][16]) and (i >> 1) * 4096 + (i & 1) * 512 in a GEMM 2 stage ([k slab][64 rows][16]) when a
  // wave covers more than one k slab (PP = 4), i * 512 otherwise
#define FUSED_DMA_PIECE(t, pl, i)                                                                            \
  {                                                                                                          \
    const int u_ = (t) - NS1;                                                                                \
    const int sbase = (t) < NS1 ? SPS * (t) * 4096 * 2 : ((KPS * (u_ % SPQ)) * 4096 + 64 * (u_ / SPQ) * 16) * 2; \
    const int src_off = ((t) < NS1 || PP == 2) ? (i) * 512 : ((i) >> 1) * 4096 + ((i) & 1) * 512;           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                \
        (t) < NS1 ? rs_c : rs_o,                                                                             \
        (__attribute__((address_space(3))) void*)(wbuf + ((t) & 1) * BUF + (pl) * PLANE + (PP * wave + (i)) * 512), 16, \
        ((t) < NS1 ? dvoff1 : dvoff2) * 2, sbase + (pl) * plane_bytes + src_off * 2, 0, 0);                  \
  }
#define FUSED_DMA_STAGE(t)                                                                                   \
  {                                                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                                         \
      _Pragma("unroll") for (int i = 0; i < PP; ++i) FUSED_DMA_PIECE(t, pl, i)                               \
  }
#define FUSED_PIPE_BEGIN(t)                                         \
  if constexpr (!kNoSync) {                                         \
    if ((t) < LAST_T && !(kLate16 && (t) + 1 == NS1)) {             \
      FUSED_DMA_STAGE((t) + 1)                                      \
      __builtin_amdgcn_sched_barrier(0);                            \
    } else if (has_next) {                                          \
      FUSED_DMA_STAGE(FIRST_T)                                      \
      __builtin_amdgcn_sched_barrier(0);                            \
    }                                                               \
  }
  // end of iteration t: the requests of stage t+1 must have landed before the barrier.  In GEMM 1 with the e
  // stream on the DMA queue as well, the two youngest requests are slab t+2 of e, which may stay in flight.
#define FUSED_PIPE_END(t)                                                        \
  if (!kNoSync && (t) < LAST_T) {                                                \
    if constexpr (!kNoWait) {                                                    \
      __builtin_amdgcn_s_waitcnt(0x0F70);                            /* vmcnt(0) */ \
    }                                                                            \
    if constexpr (!kNoBar) __syncthreads();                                      \
  }
#define FUSED_STAMP(k) \
  if constexpr ((ABL & 16) != 0) stamp[k] = __builtin_amdgcn_s_memtime();

  // LOG2E DOMAIN (round 5).  The pre-activation of the gate, e' = A h[j] + B h[i] + C e + b_C, is carried as e' log2(e): the node-row
  // linear delivers the A / B rows already multiplied by log2(e) with b_C folded into the A rows (weights.py: node4 "fused" bias /
  // column-scale vectors), and the accumulator of GEMM 1 is brought in by ONE multiply-add with inv1 = log2(e) 2^-kc 2^-kx.  Then
  //   sigmoid(e')  = 1 / (1 + exp2(-e' log2(e)))                      no multiply in front of v_exp_f32 (the negation is a source modifier),
  //   LayerNorm_e  is invariant under the positive factor when eps is scaled by its square (kEps1 below),
  // i.e. 2 vector instructions per value fewer in the gate (the bias add and the sigmoid's pre-multiply) and no b_C reads from LDS.
  // The activation of GEMM 2 is produced the same way: z' = z log2(e) straight from LayerNorm_o (g_o, b_o are multiplied by log2(e)
  // when they are written to LDS), a' = z' / ((1 + exp2(-z')) 2^-ka) = SiLU(z) log2(e) 2^ka with the operand scale 2^ka inside the
  // multiply-add that forms the denominator; the factor 1 / log2(e) is part of inv2 (the accumulator scale of GEMM 2).
  // scales (T::kScaled) = {log2(e) 2^-kc, 2^-(ko+ka) / log2(e), log2(e), 2^-ka}; unscaled planes (bf16): {log2(e), 1 / log2(e), log2(e), 1}.
  constexpr float kLog2e = 1.4426950408889634f;
  constexpr float kEps1 = 1e-5f * kLog2e * kLog2e;      // LayerNorm_e on e' log2(e)
  float inv2 = 1.0f / kLog2e, gmul = kLog2e, s_den = 1.0f, inv_c = kLog2e;
  if constexpr (T::kScaled) {
    inv_c = scales[0];
    inv2 = scales[1];
    gmul = scales[2];
    s_den = scales[3];
  }
  // ---- once per workgroup: requests of its first tile, layer parameters -> LDS (thread = feature) ------------------
  if (wt < wt_end) {
    if constexpr (!L0) {
      ring_fill(wt * WAVES + wave);
      if constexpr (T::kScaled) tmax_cur = etmax_in[wt * WAVES + wave];      // (wave uniform: scalar load)
    }
    if constexpr (!(kLate16 && L0)) { FUSED_DMA_STAGE(FIRST_T) }
  }
  if (tid < H) {
    prm[P_GE * H + tid] = g_e[tid];
    prm[P_BE * H + tid] = b_e[tid];
    prm[P_T * H + tid] = time_on_edge ? tbias[tid] : 0.0f;
    prm[P_GO * H + tid] = g_o[tid] * gmul;      // LN_o output arrives as z log2(e)
    prm[P_BO * H + tid] = b_o[tid] * gmul;
    prm[P_BOUT * H + tid] = b_out[tid];
    if constexpr (L0) {
      prm[P_TAB0 * H + tid] = l0_table[tid];
      prm[P_TAB1 * H + tid] = l0_table[H + tid];
      prm[P_CE0 * H + tid] = l0_table[2 * H + tid] * kLog2e;      // (C e_in, carried in the log2(e) domain like the accumulators)
      prm[P_CE1 * H + tid] = l0_table[3 * H + tid] * kLog2e;
    }
  }
  const int a_off = wslot(l31, hh);   // entry = 32 nb + l31 : (entry >> 3) & 1 == (l31 >> 3) & 1

#pragma unroll 1
  for (; wt < wt_end; wt += wt_step) {
  const bool has_next = kPersist && wt + wt_step < wt_end;      // (wave uniform)
  const int tile = wt * WAVES + wave;
  const int s_raw = tile * 32 + l31;
  const bool valid = s_raw < n_edges;
  const bool tile_full = (tile + 1) * 32 <= n_edges;      // (wave uniform)
  const int s = valid ? s_raw : n_edges - 1;   // lanes past the end redo the last edge and are masked out
  // e is stored TILED ("MFMA native", see edge_tiled_offset in kernels.h): per 32-edge tile the 1 KiB that one
  // wave instruction touches is contiguous, so every access below is a fully coalesced 1 KiB transaction
  // Addressing is (wave-uniform 64-bit base + compile-time constant) + 32-bit lane offset so that the loads use
  // the scalar-base form; per-lane 64-bit pointers with large constant offsets cost a VGPR pair per address.
  float* const etile = e + (long long)tile * (32 * H);   // + slab * 512 + i * 256 + loff
  const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(etile, 0, 32 * H * 4, 0x00020000);
  auto ld_e = [&](int off, bool buf = false) -> v4f {       // off: float offset inside the tile (a constant at every call)
    if (buf) {
      const v4u_ r = __builtin_amdgcn_raw_buffer_load_b128(rs_e, loff_b, off * 4, kNtRing ? 2 : 0);
      return __builtin_bit_cast(v4f, r);
    } else {
      const float* p = etile + off + loff;
      if constexpr (kNtRes) return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
      else return *reinterpret_cast<const v4f*>(p);
    }
  };
  auto st_e = [&](int off, v4f v) {
    float* p = etile + off + loff;
    if constexpr (kNtSt) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
    else *reinterpret_cast<v4f*>(p) = v;
  };
  float sx = 1.0f, inv1 = inv_c;      // accumulator of GEMM 1 -> e' log2(e): log2(e) (2^-kc 2^-kx)
  if constexpr (T::kScaled && !L0) {
    float invx;
    sx = pow2_scale_for(tmax_cur, invx);      // (wave uniform: scalar arithmetic)
    inv1 = inv_c * invx;
  }
  float tmx = 0.0f;      // max |e_new| over this lane's share of the tile (kScaled: becomes etmax_out[tile])
  // phase timestamps live in SGPRs and are written once at the end (ABL & 16 only)
  unsigned long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  FUSED_STAMP(0)

  v16f acc1[8];
  // top of a tile: its first weight stage (requested during the previous tile's last stage, or above), the ring slabs and
  // - first tile - the parameters have landed
  // layer 0: this lane's table row (float offset into prm), rule of table_rows_tiled_kernel
  int l0_row = P_TAB0 * H;
  if constexpr (L0) {
    if (l0_x != nullptr && l0_x[l0_perm ? l0_perm[s] : s] > 0.5f) l0_row = P_TAB1 * H;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  const int j = col[s];
  const int i_node = row[s];
  const float* nj = node4 + (long long)j * 4 * H;       // rows U | V | A | B
  const float* ni = node4 + (long long)i_node * 4 * H;

  FUSED_STAMP(1)
  if constexpr (L0) {      // C e_in of this lane's features, from the row that belongs to its input row
    const int ce_row = l0_row + (P_CE0 - P_TAB0) * H;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4f c = *reinterpret_cast<const v4f*>(prm + ce_row + 32 * nb + 8 * g + 4 * hh);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc1[nb][4 * g + q] = c[q];
      }
  } else {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[nb][r] = 0.0f;
  }

  // ================================ GEMM 1 ==========================================================
  // OPT bit 5: two stages of cover for the e stream.  The slab of stage t is split BEFORE this stage's weight requests
  // are issued and the ring slot is refilled AFTER them, so the two youngest requests of a stage are its e loads; the
  // stage then ends with vmcnt(2) instead of vmcnt(0): the weight pieces have landed (requests complete in order), the
  // e loads stay in flight across the barrier and are waited for at the end of the NEXT stage.
  constexpr bool kDeepE = (OPT & 32) != 0;
  // OPT bit 21 (experiment): no scheduling fences around the MFMA triples of the GEMM loops - the compiler places the fragment reads and
  // the operand splits between the MFMAs as its scheduler sees fit
  constexpr bool kFreeSched = (OPT & 2097152) != 0;
  static_assert(!kDeepE || (SPS == 1 && RING == 2), "OPT bit 5 is written for the 16 KiB stages");
#pragma unroll
  for (int t = 0; t < (L0 ? 0 : NS1); ++t) {
    if constexpr (!kDeepE) { FUSED_PIPE_BEGIN(t) }
    // B operands of the slab(s) of this stage
    frag xh[SPS], xl[SPS];
#pragma unroll
    for (int sub = 0; sub < SPS; ++sub) {
      const int ks = SPS * t + sub;
      v4f c0, c1;
      {
        c0 = er[ks % RING][0];
        c1 = er[ks % RING][1];
        if (!kDeepE && !kNoE && ks + RING < 16) {
          er[ks % RING][0] = ld_e((ks + RING) * 512, kBufRing);
          er[ks % RING][1] = ld_e(((ks + RING) * 512 + 256), kBufRing);
        }
      }
      if constexpr (T::kScaled) {
        c0 = c0 * sx;
        c1 = c1 * sx;
      }
      const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      split8<T>(xs, xh[sub], xl[sub]);
    }
    if constexpr (kDeepE) {
      __builtin_amdgcn_sched_barrier(0);
      FUSED_PIPE_BEGIN(t)
      if (!kNoE && t + RING < 16) {
        er[t % RING][0] = ld_e((t + RING) * 512, kBufRing);
        er[t % RING][1] = ld_e(((t + RING) * 512 + 256), kBufRing);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // 8 SPS weight blocks (bi = sub * 8 + nb), A fragments read from LDS two blocks ahead of their MFMAs
    const unsigned short* wb = wbuf + (t & 1) * BUF + a_off;
    frag fh[4], fl[4];
#define FUSED_FRAG1(bi, slot)                                                                        \
  {                                                                                                  \
    fh[slot] = *reinterpret_cast<const frag*>(wb + ((bi) >> 3) * 256 * 16 + ((bi) & 7) * 32 * 16);   \
    fl[slot] = *reinterpret_cast<const frag*>(wb + PLANE + ((bi) >> 3) * 256 * 16 + ((bi) & 7) * 32 * 16); \
  }
    if constexpr ((OPT & 2) != 0) {
      // pairs of blocks: fragments of pair bp + 1 are requested before the six MFMAs of pair bp, whose two accumulator
      // chains alternate (per accumulator the order of the products is unchanged: bit-identical results)
      FUSED_FRAG1(0, 0)
      FUSED_FRAG1(1, 1)
#pragma unroll
      for (int bp = 0; bp < 4 * SPS; ++bp) {
        if (bp + 1 < 4 * SPS) {
          FUSED_FRAG1(2 * bp + 2, 2 * ((bp + 1) & 1))
          FUSED_FRAG1(2 * bp + 3, 2 * ((bp + 1) & 1) + 1)
        }
        __builtin_amdgcn_sched_barrier(0);
        const int s0 = 2 * (bp & 1), s1 = s0 + 1, n0 = (2 * bp) & 7, n1 = n0 + 1, sub = (2 * bp) >> 3;
        acc1[n0] = T::mfma(fl[s0], xh[sub], acc1[n0]);
        acc1[n1] = T::mfma(fl[s1], xh[sub], acc1[n1]);
        acc1[n0] = T::mfma(fh[s0], xl[sub], acc1[n0]);
        acc1[n1] = T::mfma(fh[s1], xl[sub], acc1[n1]);
        acc1[n0] = T::mfma(fh[s0], xh[sub], acc1[n0]);
        acc1[n1] = T::mfma(fh[s1], xh[sub], acc1[n1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      FUSED_FRAG1(0, 0)
      FUSED_FRAG1(1, 1)
#pragma unroll
      for (int bi = 0; bi < 8 * SPS; ++bi) {
        if (bi + 2 < 8 * SPS) FUSED_FRAG1(bi + 2, (bi + 2) % 3)
        if constexpr (!kFreeSched) __builtin_amdgcn_sched_barrier(0);
        const int nb = bi & 7, sub = bi >> 3;
        acc1[nb] = T::mfma(fl[bi % 3], xh[sub], acc1[nb]);
        acc1[nb] = T::mfma(fh[bi % 3], xl[sub], acc1[nb]);
        acc1[nb] = T::mfma(fh[bi % 3], xh[sub], acc1[nb]);
        if constexpr (!kFreeSched) __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef FUSED_FRAG1
    if constexpr (kDeepE) {
      if (!kNoE && t + RING < 16) {
        if constexpr (!kNoWait) __builtin_amdgcn_s_waitcnt(0x0F72);      // vmcnt(2): everything but this stage's two e loads
        if constexpr (!kNoBar) __syncthreads();
      } else {
        FUSED_PIPE_END(t)
      }
    } else {
      FUSED_PIPE_END(t)
    }
    if (t == 0) { FUSED_STAMP(2) }
    if (t == NS1 / 2 - 1) { FUSED_STAMP(3) }
  }

  FUSED_STAMP(4)
  // OPT bit 9: the latency-bound phases (gathers, gate, neighbour sum, LayerNorms) run at raised issue priority (s_setprio 3):
  // their dependent VALU chains no longer queue behind the co-resident wave's MFMA issue.  +1.2 % on the step; raising
  // the prologue and GEMM 2's output phases as well measured 0.5 % less, raising the GEMM phases instead (round 1) lost 1 %.
  if constexpr ((OPT & 512) != 0) __builtin_amdgcn_s_setprio(3);
  // ================================ epilogue 1 =======================================================
  // quad (nb, g): features fb = 32 nb + 8 g + 4 hh + 0..3 of edge s, accumulator registers 4g..4g+3.
  // Neighbour-table rows: full-line gathers through LDS (OPT bit 14, production) or register gathers one batch (= 2 quads) ahead.
  // OPT bit 6: the four LayerNorm reductions (sum, centred sum of squares, twice) run as FOUR interleaved partial sums per
  // lane instead of one 128-term serial chain each; same terms, different summation order (fp32 rounding only)
  static_assert((OPT & 64) != 0, "OPT bit 6 is part of every kept instantiation (the serial-sum form was removed in round 5)");
  // OPT bit 11: the element-wise arithmetic of the gate, the two LayerNorms and the activation is written on register PAIRS
  // (elements 2p, 2p+1 of an accumulator tuple, .xy / .zw of the parameter vectors) so that it maps to v_pk_add / v_pk_mul /
  // v_pk_fma_f32 without the v_mov pairs hipcc's own vectoriser needed for the pairs it chose; element for element the same
  // operations in the same order as the scalar code (requires bit 6: the partial sums are the pairs' running sums)
  static_assert((OPT & 2048) != 0, "OPT bit 11 is part of every kept instantiation (the scalar form was removed in round 5)");
  // OPT bit 17: fast path of the neighbour sum for tiles that hold ONE centre node (see FUSED_AGG_ROUND); bit-identical
  constexpr bool kAggFast = (OPT & 131072) != 0;
  // (round 5, with bit 17) the general path tests for a segment start only inside the 8-row groups that hold one
  constexpr bool kAggQuarter = kAggFast;
  // OPT bit 19 (a SEMANTIC switch, its own instantiations: kinds 8, 9, 11 of launch_fused_kind): aggregation = "max"
  // (gnn_encoder.py:172-173,187-188) - the per-segment pieces part / direct hold the element-wise MAXIMUM of the gated messages
  // instead of their sum (node_finalize_kernel then combines the pieces of a node by maximum); the pad lanes of a launch's last
  // tile contribute -inf instead of 0.  Everything else of the layer is unchanged.
  constexpr bool kAggMax = (OPT & 524288) != 0;
  const float agg_neutral = kAggMax ? -__builtin_inff() : 0.0f;
  float s1 = 0.0f;
  v2f s1k[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
  // segment structure of the tile: bit k of bnd = edge k starts a new centre node (wave uniform)
  const int i_prev = __shfl_up(i_node, 1, 64);
  const unsigned bnd = (unsigned)__ballot(l31 > 0 && i_node != i_prev);
  const int first_end = bnd ? __builtin_ctz(bnd) : 32;
  float* part0 = part + ((long long)tile * 2 + 0) * H;
  float* part1 = part + ((long long)tile * 2 + 1) * H;

  constexpr int GD = 2;      // register gathers (n_nodes >= 2^20 instantiations): one batch ahead of their use, two ring slots
  v4f ga[GD][2][3];
#define FUSED_GATHER(b, buf)                                                          \
  {                                                                                   \
    _Pragma("unroll") for (int q2 = 0; q2 < 2; ++q2) {                                \
      const int fb_ = 32 * ((b) >> 1) + 8 * (2 * ((b) & 1) + q2) + 4 * hh;            \
      if constexpr (!(ablate & 1) && !(ablate & 256)) {                               \
        ga[buf][q2][0] = *reinterpret_cast<const v4f*>(nj + 2 * H + fb_);             \
        if constexpr (TAIL != 1) ga[buf][q2][2] = *reinterpret_cast<const v4f*>(nj + H + fb_); \
      } else {                                                                        \
        ga[buf][q2][0] = ga[buf][q2][2] = v4f{0.f, 0.f, 0.f, 0.f};                    \
      }                                                                               \
      if constexpr (!(ablate & 1) && !(ablate & 128)) {                               \
        ga[buf][q2][1] = *reinterpret_cast<const v4f*>(ni + 3 * H + fb_);             \
      } else {                                                                        \
        ga[buf][q2][1] = v4f{0.f, 0.f, 0.f, 0.f};                                     \
      }                                                                               \
    }                                                                                 \
  }
  // one aggregation round: the gated messages of 64 features (two blocks) of all 32 edges sit in the wave's scratch
#define FUSED_AGG_ROUND(rnd)                                                                                                 \
      __builtin_amdgcn_wave_barrier();                                                                                       \
      if constexpr (!(ablate & 2)) {                                                                                         \
        const int f = 64 * (rnd) + lane;                                                                                       \
        /* two halves of 16 rows: 32 values in flight at once is the register peak of the kernel (128 accumulators + the */  \
        /* gather buffers are live here) and made the compiler spill accumulators */                                         \
        float accv = agg_neutral;                                                                                            \
        if (kAggFast && bnd == 0) {                                                                                          \
          /* (wave uniform) ONE centre node in the whole tile - 68 % of the tiles at K = 100: plain column sums in the */    \
          /* same order, without the per-row boundary tests (5 scalar instructions + a taken branch per row) */              \
_Pragma("unroll")                                                                                                         \
          for (int half = 0; half < 2; ++half) {                                                                             \
            float v[16];                                                                                                     \
_Pragma("unroll")                                                                                                         \
            for (int k = 0; k < 16; ++k) v[k] = scr[(16 * half + k) * SCR_STRIDE + lane];                                    \
_Pragma("unroll")                                                                                                         \
            for (int kk = 0; kk < 16; ++kk) { if constexpr (kAggMax) accv = __builtin_fmaxf(accv, v[kk]); else accv += v[kk]; } \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
          }                                                                                                                  \
          part0[f] = accv;                                                                                                   \
        } else {                                                                                                             \
_Pragma("unroll")                                                                                                         \
        for (int half = 0; half < 2; ++half) {                                                                               \
          float v[16];                                                                                                       \
_Pragma("unroll")                                                                                                         \
          for (int k = 0; k < 16; ++k) v[k] = scr[(16 * half + k) * SCR_STRIDE + lane];                                      \
          /* (round 5) per group of 8 rows: a group that holds no segment start (wave uniform; with K = 100 a tile has at most */ \
          /* one, so three of its four groups) adds its rows without the per-row tests - same order, bit-identical */         \
_Pragma("unroll")                                                                                                         \
          for (int q8 = 0; q8 < 2; ++q8) {                                                                                   \
            if (kAggQuarter && ((bnd >> (16 * half + 8 * q8)) & 0xFFu) == 0) {                                               \
_Pragma("unroll")                                                                                                         \
              for (int kk = 8 * q8; kk < 8 * q8 + 8; ++kk) { if constexpr (kAggMax) accv = __builtin_fmaxf(accv, v[kk]); else accv += v[kk]; } \
            } else {                                                                                                         \
_Pragma("unroll")                                                                                                         \
          for (int kk = 8 * q8; kk < 8 * q8 + 8; ++kk) {                                                                     \
            const int k = 16 * half + kk;                                                                                    \
            if (k > 0 && ((bnd >> k) & 1u)) {                      /* wave-uniform branch */                                 \
              const int node = __builtin_amdgcn_readlane(i_node, k - 1);                                                     \
              float* dst = (k == first_end) ? part0 : direct + (long long)node * H;                                          \
              dst[f] = accv;                                                                                                 \
              accv = agg_neutral;                                                                                            \
            }                                                                                                                \
            if constexpr (kAggMax) accv = __builtin_fmaxf(accv, v[kk]); else accv += v[kk];                                  \
          }                                                                                                                  \
            }                                                                                                                \
          }                                                                                                                  \
          __builtin_amdgcn_sched_barrier(0);                                                                                 \
        }                                                                                                                    \
        float* dst = (first_end == 32) ? part0 : part1;                                                                      \
        dst[f] = accv;                                                                                                       \
        }                                                                                                                    \
      }                                                                                                                      \
      __builtin_amdgcn_wave_barrier();
  // OPT bit 14 (experiment): A h[j] / V h[j] by FULL-LINE gathers.  With lane = edge every gather instruction touches 32 rows x
  // 32 B: 2,048 line look-ups per tile for 512 distinct lines.  Here a (table, block nb) unit - 32 rows x 128 B = the 32 features
  // of block nb - is fetched by four LDS-DMA instructions (lane L: row 8 p + L / 8, 16-byte chunk (L % 8) ^ swz(row), i.e. eight
  // full lines per instruction) into the wave's share of weight buffer 1, which is idle between the end of GEMM 1 and the wave's
  // own request of stage 17 (its two 2-KB piece areas: rows 0-15 | 16-31), and read back as four ds_read_b128 per lane
  // (chunk 2 g + hh at position (2 g + hh) ^ swz(edge): conflict free with swz(r) = ((r >> 1) & 3) | ((r >> 4) & 1) << 2).
  // One buffer, the two tables alternate: A(nb) is read, V(nb) requested, the gate's e' / sigmoid computed, V(nb) read,
  // A(nb + 1) requested, the messages formed.  Every wait is vmcnt(0): stores share the counter on gfx9.
  constexpr bool kFL = (OPT & 16384) != 0;
  static_assert(!kFL || (ablate & 0x30F) == 0, "OPT bit 14 is written for the production arithmetic");      // (bits 5-7: GEMM 2 output path / B h[i] loads off - traffic attribution, round 6)
  // ABL 1024 / 2048 (profiling library, wrong results): the full-line gather requests are issued but never waited for / not
  // issued at all - what the waits and what the issue of the 64 LDS-DMA pieces cost in the gather phase
  // OPT bit 15 (with bit 14): TWO units - A in the wave's share of buffer 1, V in its share of buffer 0 - so that the next block
  // of a table is requested as soon as the current one has been read (a whole block of cover instead of half).  Buffer 0 is
  // free because GEMM 2's first weight stage is then requested AFTER the gather phase (kLate16; the LayerNorm phase covers it)
  // and met by one extra workgroup barrier in front of GEMM 2.  Waits are counted: at every wait the operations allowed to stay
  // in flight are the NEWEST loads (the other table's block + the B h[i] rows), and loads return in order, so an old
  // request cannot be pending when the count is reached - whatever the neighbour-sum stores (same counter) do.
  constexpr bool kFL2 = kFL && (OPT & 32768) != 0;
  static_assert(!kFL2 || !kPersist, "bits 12 and 15 are not combined");
  static_assert(!kAggMax || kFL, "OPT bit 19 (max aggregation) is written for the production gather path");
  if constexpr (kFL) {
    const int rsel = lane >> 3, cc = lane & 7;
    auto swz = [](int r) { return ((r >> 1) & 3) | (((r >> 4) & 1) << 2); };
    unsigned src_off[4];
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
      const int r = 8 * p4 + rsel;
      const int jr = __shfl(j, r, 64);
      src_off[p4] = (unsigned)jr * (unsigned)(4 * H * 4) + (unsigned)((cc ^ swz(r)) * 16);
    }
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(node4), 0, 0xffffffff, 0x00020000);
    // unit u: rows 0-15 in the plane-0 piece area, rows 16-31 in the plane-1 piece area of this wave, buffer 1 - u
    unsigned short* const ua0 = wbuf + BUF + (PP * wave) * 512;
    unsigned short* const ua1 = wbuf + BUF + PLANE + (PP * wave) * 512;
    unsigned short* const ub0 = wbuf + (PP * wave) * 512;
    unsigned short* const ub1 = wbuf + PLANE + (PP * wave) * 512;
    const int rd_row = ((l31 & 16) ? PLANE * 2 : 0) + (l31 & 15) * 128;      // byte offset of this lane's row inside a unit
    const unsigned char* const rd_a = reinterpret_cast<const unsigned char*>(ua0) + rd_row;
    const unsigned char* const rd_b = reinterpret_cast<const unsigned char*>(ub0) + rd_row;
    int rd_pos[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rd_pos[g] = ((2 * g + hh) ^ swz(l31)) * 16;
    // table: 1 = V (float offset H), 2 = A (float offset 2 H) of the node4 row; unit 0 = buffer 1 areas, 1 = buffer 0 areas
#define FUSED_FL_REQUEST(table, nb_, unit)                                                                                  \
  if constexpr ((ABL & 2048) == 0) {      /* (ABL 2048: timing only - no gather requests at all: stale LDS is read) */     \
    _Pragma("unroll") for (int p4 = 0; p4 < 4; ++p4)                                                                      \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_n,                                                                      \
          (__attribute__((address_space(3))) void*)(((unit) ? ((p4 >> 1) ? ub1 : ub0) : ((p4 >> 1) ? ua1 : ua0)) + (p4 & 1) * 512), \
          16, src_off[p4], (table) * H * 4 + (nb_) * 128, 0, 0);                                                          \
  }
#define FUSED_FL_WAIT(n)                                                          \
  if constexpr ((ABL & 1024) == 0) {      /* (ABL 1024: timing only - the gather requests are never waited for) */ \
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14));         \
  }                                                                               \
  __builtin_amdgcn_sched_barrier(0);                                              \
  asm volatile("" ::: "memory");
#define FUSED_FL_READ(dst, unit)                                                                      \
  {                                                                                                   \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                     \
      dst[g] = *reinterpret_cast<const v4f*>(((unit) ? rd_b : rd_a) + rd_pos[g]);                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
    // B h[i] rows through the same buffer resource (MUBUF like the LDS-DMA requests: one in-order load queue for the counted waits)
    v4f bh_[2][4];
    const int b_voff = i_node * (4 * H * 4) + hh * 16;
#define FUSED_FL_B(nb_, buf)                                                                          \
  {                                                                                                   \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                   \
      if constexpr ((ablate & 128) == 0)                                                              \
        bh_[buf][g] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_n, b_voff, (3 * H + 32 * (nb_) + 8 * g) * 4, 0)); \
      else bh_[buf][g] = v4f{0.f, 0.f, 0.f, 0.f};      /* (ABL 128, timing / traffic attribution only: no B h[i] loads) */ \
    }                                                                                                 \
  }
    FUSED_FL_B(0, 0)
    FUSED_FL_REQUEST(2, 0, 0)
    if constexpr (kFL2 && TAIL != 1) { FUSED_FL_REQUEST(1, 0, 1) }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int nq = nb & 1;
      if (nb + 1 < 8) {
        if (((nb + 1) & 1) == 0) FUSED_FL_B(nb + 1, 0) else FUSED_FL_B(nb + 1, 1)
        if constexpr (kFL2 && TAIL == 1) {      // A alternates between the two units, one block ahead
          if (((nb + 1) & 1) == 0) FUSED_FL_REQUEST(2, nb + 1, 0) else FUSED_FL_REQUEST(2, nb + 1, 1)
        }
      }
      v4f ah_q[4], vh_q[4];
      v2f sg_q[4][2];
      if constexpr (kFL2) {
        if (nb + 1 < 8) { FUSED_FL_WAIT(8) }           // in flight: the other table's block (V(nb) | A(nb+1)) + B(nb+1)
        else if (TAIL != 1) { FUSED_FL_WAIT(4) }       // V(7)
        else { FUSED_FL_WAIT(0) }
      } else {
        FUSED_FL_WAIT(0)
      }
      if constexpr (kFL2 && TAIL == 1) {
        if ((nb & 1) == 0) FUSED_FL_READ(ah_q, 0) else FUSED_FL_READ(ah_q, 1)
      } else {
        FUSED_FL_READ(ah_q, 0)
      }
      if constexpr (kFL2) {
        if constexpr (TAIL != 1) {
          if (nb + 1 < 8) { FUSED_FL_REQUEST(2, nb + 1, 0) }
        }
      } else {
        if constexpr (TAIL != 1) { FUSED_FL_REQUEST(1, nb, 0) }
        else if (nb + 1 < 8) { FUSED_FL_REQUEST(2, nb + 1, 0) }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int r = 4 * g + 2 * h2;
          // e' log2(e) = (C e) inv1 + (A h[j] + b_A + b_C) log2(e) + B h[i] log2(e)      (layer 0: the accumulators hold C e_in log2(e))
          v2f ev;
          if constexpr (!L0) ev = DIFUSCO_PAIR(acc1[nb], r) * v2f{inv1, inv1} + DIFUSCO_PAIR(ah_q[g], 2 * h2);
          else ev = DIFUSCO_PAIR(acc1[nb], r) + DIFUSCO_PAIR(ah_q[g], 2 * h2);
          ev = ev + DIFUSCO_PAIR(bh_[nb & 1][g], 2 * h2);
          acc1[nb][r] = ev[0];
          acc1[nb][r + 1] = ev[1];
          s1k[h2] += ev;
          if constexpr (TAIL != 1) sg_q[g][h2] = sigmoid2_log2e(ev);
        }
      }
      if constexpr (TAIL != 1) {
        if constexpr (kFL2) {
          if (nb + 1 < 8) { FUSED_FL_WAIT(8) }         // in flight: B(nb+1), A(nb+1)
          else { FUSED_FL_WAIT(0) }
          FUSED_FL_READ(vh_q, 1)
        } else {
          FUSED_FL_WAIT(0)
          FUSED_FL_READ(vh_q, 0)
          if (nb + 1 < 8) { FUSED_FL_REQUEST(2, nb + 1, 0) }
        }
        if (tile_full) {      // (wave uniform) every tile but the last of a launch: no per-element selects
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            v4f m;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const v2f sg = sg_q[g][h2] * DIFUSCO_PAIR(vh_q[g], 2 * h2);
              m[2 * h2] = sg[0];
              m[2 * h2 + 1] = sg[1];
            }
            *reinterpret_cast<v4f*>(scr + l31 * SCR_STRIDE + nq * 32 + 8 * g + 4 * hh) = m;
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            v4f m;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const v2f sg = sg_q[g][h2] * DIFUSCO_PAIR(vh_q[g], 2 * h2);
              m[2 * h2] = valid ? sg[0] : agg_neutral;
              m[2 * h2 + 1] = valid ? sg[1] : agg_neutral;
            }
            *reinterpret_cast<v4f*>(scr + l31 * SCR_STRIDE + nq * 32 + 8 * g + 4 * hh) = m;
          }
        }
        if (nq == 1) { FUSED_AGG_ROUND(nb >> 1) }
        // (after the stores of the round: the newest operations at the next wait must be loads)
        if constexpr (kFL2) {
          if (nb + 1 < 8) { FUSED_FL_REQUEST(1, nb + 1, 1) }
        }
      }
    }
    // GEMM 2's first weight stage, requested now that buffer 0 is free again (its landing is met in front of GEMM 2)
    if constexpr (kFL2 && TAIL != 2) { FUSED_DMA_STAGE(NS1) }
#undef FUSED_FL_REQUEST
#undef FUSED_FL_WAIT
#undef FUSED_FL_READ
#undef FUSED_FL_B
  } else {
  FUSED_GATHER(0, 0)
#pragma unroll
  for (int b = 0; b < 16; ++b) {             // batch b: block nb = b >> 1, quads g = 2 (b & 1) + {0, 1}
    if (b + 1 < 16) {
      if (((b + 1) & 1) == 0) FUSED_GATHER(b + 1, 0) else FUSED_GATHER(b + 1, 1)
    }
    if constexpr ((OPT & 256) != 0) __builtin_amdgcn_sched_barrier(0);
    const int nb = b >> 1, nq = nb & 1;
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int g = 2 * (b & 1) + q2;
      const int fb = 32 * nb + 8 * g + 4 * hh;
      const v4f ah = ga[b % GD][q2][0], bh = ga[b % GD][q2][1], vh = ga[b % GD][q2][2];
      v4f m;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int r = 4 * g + 2 * h2;
        v2f ev;      // (the arithmetic of the full-line branch above, operation for operation: bit-identical results)
        if constexpr (!L0) ev = DIFUSCO_PAIR(acc1[nb], r) * v2f{inv1, inv1} + DIFUSCO_PAIR(ah, 2 * h2);
        else ev = DIFUSCO_PAIR(acc1[nb], r) + DIFUSCO_PAIR(ah, 2 * h2);
        ev = ev + DIFUSCO_PAIR(bh, 2 * h2);
        acc1[nb][r] = ev[0];
        acc1[nb][r + 1] = ev[1];
        s1k[h2] += ev;
        if constexpr (TAIL != 1) {
          const v2f sg = sigmoid2_log2e(ev) * DIFUSCO_PAIR(vh, 2 * h2);
          m[2 * h2] = valid ? sg[0] : 0.0f;
          m[2 * h2 + 1] = valid ? sg[1] : 0.0f;
        }
      }
      if constexpr (TAIL != 1) *reinterpret_cast<v4f*>(scr + l31 * SCR_STRIDE + nq * 32 + 8 * g + 4 * hh) = m;
    }
    if (TAIL != 1 && (b & 3) == 3) {
      // 64 features (blocks 2 rnd, 2 rnd + 1) of all 32 edges are in the scratch: segmented column sums,
      // lane = feature 64 rnd + lane.  The first segment of a tile may continue from the previous tile and
      // the last into the next one (part[tile][0|1]); inner segments are complete (direct[node]).
      FUSED_AGG_ROUND(b >> 2)
    }
  }
  }      // !kFL
#undef FUSED_GATHER
#undef FUSED_AGG_ROUND
  FUSED_STAMP(5)
  // requests of the workgroup's next tile that would otherwise be exposed at its top: the first ring slabs of e and the
  // tile's scale.  The ring registers are dead after GEMM 1; issued here (MIS last layer) / behind the last MFMA of GEMM 2,
  // they are covered by the last output phase.
#define FUSED_NEXT_TILE_REQUESTS                                                       \
  if constexpr (!L0) {                                                                 \
    if (has_next) {                                                                    \
      const int tile_n = (wt + wt_step) * WAVES + wave;                                \
      ring_fill(tile_n);                                                               \
      if constexpr (T::kScaled) tmax_cur = etmax_in[tile_n];                           \
    }                                                                                  \
  }
  if constexpr (TAIL == 2) {      // the edge output of this layer is never read
    FUSED_NEXT_TILE_REQUESTS
    continue;
  }

  // LayerNorm_e (two pass on registers), ReLU, + t, LayerNorm_o, SiLU.  Element-wise arithmetic on register pairs (v2f): with the
  // packed-fp32 target feature off (build.py) every pair operation becomes two plain instructions on adjacent registers.
  constexpr float inv_h = 1.0f / 256.0f;
  constexpr bool skip_math = (ablate & 4) != 0;
  s1 = (s1k[0][0] + s1k[0][1]) + (s1k[1][0] + s1k[1][1]);
  const float mean1 = skip_math ? 0.0f : (s1 + __shfl_xor(s1, 32, 64)) * inv_h;
  float q1;
  {
    v2f qk[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
    const v2f mean1k = {mean1, mean1};
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const v2f d = DIFUSCO_PAIR(acc1[nb], 2 * pr) - mean1k;
        acc1[nb][2 * pr] = d[0];
        acc1[nb][2 * pr + 1] = d[1];
        qk[pr & 1] += d * d;
      }
    q1 = (qk[0][0] + qk[0][1]) + (qk[1][0] + qk[1][1]);
  }
  // (the values are e' log2(e): the variance carries log2(e)^2, and so does the epsilon - the normalised value is that of e')
  const float rstd1 = __builtin_amdgcn_rsqf((q1 + __shfl_xor(q1, 32, 64)) * inv_h + kEps1);
  float s2 = 0.0f;
  v2f s2k[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
  if constexpr (!skip_math) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fb = 32 * nb + 8 * g + 4 * hh;
        const v4f ge = *reinterpret_cast<const v4f*>(prm + P_GE * H + fb);
        const v4f be = *reinterpret_cast<const v4f*>(prm + P_BE * H + fb);
        v4f tb;
        if constexpr (!NOTB) tb = *reinterpret_cast<const v4f*>(prm + P_T * H + fb);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int r = 4 * g + 2 * h2;
          v2f y = DIFUSCO_PAIR(acc1[nb], r) * v2f{rstd1, rstd1} * DIFUSCO_PAIR(ge, 2 * h2) + DIFUSCO_PAIR(be, 2 * h2);
          y = v2f{y[0] > 0.0f ? y[0] : 0.0f, y[1] > 0.0f ? y[1] : 0.0f};
          if constexpr (!NOTB) y = y + DIFUSCO_PAIR(tb, 2 * h2);      // (NOTB: the layer has no time bias on e - MIS, gnn_encoder.py:447)
          acc1[nb][r] = y[0];
          acc1[nb][r + 1] = y[1];
          s2k[h2] += y;
        }
      }
  }
  s2 = (s2k[0][0] + s2k[0][1]) + (s2k[1][0] + s2k[1][1]);
  const float mean2 = (s2 + __shfl_xor(s2, 32, 64)) * inv_h;
  float q2s;
  {
    v2f qk[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
    const v2f mean2k = {mean2, mean2};
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        const v2f d = DIFUSCO_PAIR(acc1[nb], 2 * pr) - mean2k;
        acc1[nb][2 * pr] = d[0];
        acc1[nb][2 * pr + 1] = d[1];
        qk[pr & 1] += d * d;
      }
    q2s = (qk[0][0] + qk[0][1]) + (qk[1][0] + qk[1][1]);
  }
  const float rstd2 = __builtin_amdgcn_rsqf((q2s + __shfl_xor(q2s, 32, 64)) * inv_h + 1e-5f);

  // activation -> 16-bit planes, kept in registers as the B operands of GEMM 2
  frag ah_[8][2], al_[8][2];      // [block nb][register group rg] : slab 2 nb + rg
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      float a8[8];
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int g = 2 * rg + g2;
        const int fb = 32 * nb + 8 * g + 4 * hh;
        const v4f go = *reinterpret_cast<const v4f*>(prm + P_GO * H + fb);
        const v4f bo = *reinterpret_cast<const v4f*>(prm + P_BO * H + fb);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const v2f z = DIFUSCO_PAIR(acc1[nb], 4 * g + 2 * h2) * v2f{rstd2, rstd2} * DIFUSCO_PAIR(go, 2 * h2) + DIFUSCO_PAIR(bo, 2 * h2);
          const v2f a = skip_math ? z : silu2_log2e(z, s_den);      // z log2(e) -> SiLU(z) log2(e) 2^ka
          a8[4 * g2 + 2 * h2] = a[0];
          a8[4 * g2 + 2 * h2 + 1] = a[1];
        }
      }
      split8<T>(a8, ah_[nb][rg], al_[nb][rg]);
    }

  if constexpr ((OPT & 512) != 0) __builtin_amdgcn_s_setprio(0);
  FUSED_STAMP(6)
  // ================================ GEMM 2 (four output quarters of 64 features) ======================
  constexpr bool skip_gemm2 = (ablate & 8) != 0;   // (barriers must still be executed by every wave)
  constexpr bool skip_out = (ablate & 32) != 0;    // GEMM 2 without residual read / e store
  constexpr bool skip_mm2 = (ablate & 64) != 0;    // GEMM 2 output path without its MFMAs
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    v16f acc2[2];
#pragma unroll
    for (int nbp = 0; nbp < 2; ++nbp)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[nbp][r] = 0.0f;
    v4f ein[2][4];      // residual rows of this quarter, fetched under the quarter's last 48 MFMAs
#pragma unroll
    for (int kc = 0; kc < SPQ; ++kc) {
      const int t = NS1 + qt * SPQ + kc;
      if constexpr (kLate16) {
        if (t == NS1) {      // the first stage of GEMM 2 was requested after the gather phase: every wave's pieces have landed
          __builtin_amdgcn_s_waitcnt(0x0F70);
          __syncthreads();
        }
      }
      FUSED_PIPE_BEGIN(t)
      if (kc == SPQ - 1 && !skip_gemm2 && !skip_out) {
#pragma unroll
        for (int nbp = 0; nbp < 2; ++nbp)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if constexpr (L0) ein[nbp][g] = *reinterpret_cast<const v4f*>(prm + l0_row + 64 * qt + 32 * nbp + 8 * g + 4 * hh);
            else ein[nbp][g] = ld_e(((4 * qt + 2 * nbp + (g >> 1)) * 512 + (g & 1) * 256));
          }
      }
      const unsigned short* wb = wbuf + (t & 1) * BUF + a_off;
      if constexpr (!skip_gemm2 && !skip_mm2) {
        // 2 KPS weight blocks (bi = ksl * 2 + nbp), A fragments read from LDS two blocks ahead of their MFMAs
        frag fh[4], fl[4];
#define FUSED_FRAG2(bi, slot)                                                                            \
  {                                                                                                      \
    fh[slot] = *reinterpret_cast<const frag*>(wb + (((bi) >> 1) * 64 + ((bi) & 1) * 32) * 16);           \
    fl[slot] = *reinterpret_cast<const frag*>(wb + PLANE + (((bi) >> 1) * 64 + ((bi) & 1) * 32) * 16);   \
  }
        if constexpr ((OPT & 2) != 0) {
          // the two output blocks of a k slab alternate (independent accumulators acc2[0], acc2[1])
          FUSED_FRAG2(0, 0)
          FUSED_FRAG2(1, 1)
#pragma unroll
          for (int ksl = 0; ksl < KPS; ++ksl) {
            if (ksl + 1 < KPS) {
              FUSED_FRAG2(2 * ksl + 2, 2 * ((ksl + 1) & 1))
              FUSED_FRAG2(2 * ksl + 3, 2 * ((ksl + 1) & 1) + 1)
            }
            __builtin_amdgcn_sched_barrier(0);
            const int s0 = 2 * (ksl & 1), s1 = s0 + 1;
            const int sl = KPS * kc + ksl;
            acc2[0] = T::mfma(fl[s0], ah_[sl >> 1][sl & 1], acc2[0]);
            acc2[1] = T::mfma(fl[s1], ah_[sl >> 1][sl & 1], acc2[1]);
            acc2[0] = T::mfma(fh[s0], al_[sl >> 1][sl & 1], acc2[0]);
            acc2[1] = T::mfma(fh[s1], al_[sl >> 1][sl & 1], acc2[1]);
            acc2[0] = T::mfma(fh[s0], ah_[sl >> 1][sl & 1], acc2[0]);
            acc2[1] = T::mfma(fh[s1], ah_[sl >> 1][sl & 1], acc2[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // OPT bit 20 (experiment): block-major order inside a stage - all KPS k slabs of output block 0, then of block 1: the chain
          // of MFMAs into one accumulator is 3 KPS = 12 long instead of 3 (per accumulator the k order is unchanged: bit-identical)
          constexpr bool kBlockMajor = (OPT & 1048576) != 0;
          auto frag_of = [](int bi) { return kBlockMajor ? (bi % KPS) * 2 + bi / KPS : bi; };      // fragment index ksl * 2 + nbp
          FUSED_FRAG2(frag_of(0), 0)
          FUSED_FRAG2(frag_of(1), 1)
#pragma unroll
          for (int bi = 0; bi < 2 * KPS; ++bi) {
            if (bi + 2 < 2 * KPS) FUSED_FRAG2(frag_of(bi + 2), (bi + 2) % 3)
            if constexpr (!kFreeSched) __builtin_amdgcn_sched_barrier(0);
            const int ksl = frag_of(bi) >> 1, nbp = frag_of(bi) & 1;
            const int sl = KPS * kc + ksl;        // slab of W_o = features 16 sl .. 16 sl + 15 of the activation
            acc2[nbp] = T::mfma(fl[bi % 3], ah_[sl >> 1][sl & 1], acc2[nbp]);
            acc2[nbp] = T::mfma(fh[bi % 3], al_[sl >> 1][sl & 1], acc2[nbp]);
            acc2[nbp] = T::mfma(fh[bi % 3], ah_[sl >> 1][sl & 1], acc2[nbp]);
            if constexpr (!kFreeSched) __builtin_amdgcn_sched_barrier(0);
          }
        }
#undef FUSED_FRAG2
      }
      FUSED_PIPE_END(t)
      if (t == NS1) { FUSED_STAMP(7) }
    }
    if (qt == 3) { FUSED_NEXT_TILE_REQUESTS }
    // e <- e + W_o a + b_o  for the features 64 qt + 32 nbp + 8 g + 4 hh + 0..3 of this lane's edge
    if constexpr (skip_out) {
#pragma unroll
      for (int nbp = 0; nbp < 2; ++nbp) asm volatile("" ::"v"(acc2[nbp]));   // keep the MFMAs alive
    }
    float gs[8], gq[8];      // GNP: this lane's share of the 8 groups of the quarter
#pragma unroll
    for (int u = 0; u < 8; ++u) gs[u] = gq[u] = 0.0f;
    if (valid && !skip_gemm2 && !skip_out) {
#pragma unroll
      for (int nbp = 0; nbp < 2; ++nbp)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fo = 64 * qt + 32 * nbp + 8 * g + 4 * hh;
          const v4f bo = *reinterpret_cast<const v4f*>(prm + P_BOUT * H + fo);
          v4f v;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {      // inv2 = 2^-(ko+ka) / log2(e) (unscaled planes: 1 / log2(e)): one multiply-add either way
            const v2f o = DIFUSCO_PAIR(ein[nbp][g], 2 * h2) + (DIFUSCO_PAIR(acc2[nbp], 4 * g + 2 * h2) * v2f{inv2, inv2} + DIFUSCO_PAIR(bo, 2 * h2));
            v[2 * h2] = o[0];
            v[2 * h2 + 1] = o[1];
          }
          st_e(((4 * qt + 2 * nbp + (g >> 1)) * 512 + (g & 1) * 256), v);
          if constexpr (T::kScaled && !GNP) {      // (v_max3_f32 with |.| source modifiers)
            tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
            tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
          }
          if constexpr (GNP) {
            gs[nbp * 4 + g] = (v[0] + v[1]) + (v[2] + v[3]);
            gq[nbp * 4 + g] = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          }
        }
    }
    if constexpr (GNP) {
      // all lanes again: the quarter's 16 per-lane partials (8 sums, 8 sums of squares) are summed over the 64 lanes through
      // the wave's aggregation scratch (unused by this variant): 16 writes, 16 reads and two shuffles per lane instead of
      // sixteen 6-step butterflies - those cost 4.7 k cycles per quarter, 16 % of the kernel (phase stamps, study notes section 8)
      static_assert(TAIL == 1, "the GroupNorm sums reuse the neighbour-sum scratch");
      static_assert(32 * SCR_STRIDE >= 64 * 17, "scratch too small for the 64 x 16 partials");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        scr[lane * 17 + u] = gs[u];
        scr[lane * 17 + 8 + u] = gq[u];
      }
      __builtin_amdgcn_wave_barrier();
      const int gv = lane & 15, g4 = lane >> 4;      // lane sums value gv over lanes 16 g4 .. 16 g4 + 15
      float tot = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += scr[(g4 * 16 + k) * 17 + gv];
      tot += __shfl_xor(tot, 16, 64);
      tot += __shfl_xor(tot, 32, 64);
      __builtin_amdgcn_wave_barrier();
      // gn_tile[tile][8 qt + u][sum, sum of squares]
      if (lane < 16) gn_tile[(long long)tile * 64 + qt * 16 + 2 * (gv & 7) + (gv >> 3)] = tot;
    }
    if (qt == 0) { FUSED_STAMP(8) }
  }
  if constexpr (T::kScaled && !GNP) {      // the next layer's GEMM 1 scales this tile by its max |e| (pad lanes: 0)
    if (etmax_out != nullptr) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) tmx = __builtin_fmaxf(tmx, __shfl_xor(tmx, off, 64));
      if (lane == 0) etmax_out[tile] = tmx;
    }
  }
  FUSED_STAMP(9)
  if constexpr ((ABL & 16) != 0) {
    if (dbg != nullptr && lane == 0) {
#pragma unroll
      for (int k = 0; k < 10; ++k) dbg[(long long)tile * 16 + k] = stamp[k];
    }
  }
  }      // tiles of this workgroup
#undef FUSED_NEXT_TILE_REQUESTS
#undef FUSED_STAMP
#undef FUSED_PIPE_BEGIN
#undef FUSED_PIPE_END
#undef FUSED_DMA_STAGE
#undef FUSED_DMA_PIECE
}

#ifdef DIFUSCO_PROFILING
#define FUSED_LDS_PAD g_fused_lds_pad
#define FUSED_DBG g_fused_dbg
#define FUSED_START_DELAY g_fused_start_delay
#else
#define FUSED_LDS_PAD 0
#define FUSED_DBG nullptr
#define FUSED_START_DELAY 0
#endif
#ifndef FUSED_OPT            // (-DFUSED_OPT=...: an A/B build of the production library, build.py variants)
#define FUSED_OPT 151409     // production options (OPT bits 0, 4, 5, 6, 8, 9, 10, 11, 14, 17 of the kernel)
#endif
#define FUSED_OPT_R2 3955    // round 2's production set (register gathers): what the gather / neighbour-sum ablation masks are written for
#define FUSED_NW 4          // production workgroup geometry (see fused::Geo): measured 0.97-1.00 ms vs 1.05-1.18 ms (NW = 8) per layer

template <typename T, int ABL, int NW, bool L0 = false, bool GNP = false, int TAIL = 0, int OPT = FUSED_OPT_R2, bool NOTB = false>
hipError_t launch_fused_t(float* e, const float* node4, const int* row, const int* col, int n_edges,
                                 const unsigned short* c_planes, const unsigned short* o_planes, long long plane_stride,
                                 const float* b_c, const float* g_e, const float* b_e, const float* tbias,
                                 const float* g_o, const float* b_o, const float* b_out, int time_on_edge, float* part,
                                 float* direct, hipStream_t stream, const float* l0_table, const float* l0_x,
                                 const int* l0_perm, float* gn_tile, const float* scales, const float* etmax_in,
                                 float* etmax_out) {
  static std::atomic<unsigned long long> attr_devices{0};      // per kernel instantiation: devices already configured
  {
    hipError_t er = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&edge_layer_fused_kernel<T, ABL, NW, L0, GNP, TAIL, OPT, NOTB>),
                                           160 * 1024);
    if (er != hipSuccess) return er;
  }
  constexpr int WV = fused::geo_waves(NW);
  unsigned grid = (unsigned)((n_edges + 32 * WV - 1) / (32 * WV));
  if constexpr ((OPT & 4096) != 0) {      // persistent workgroups: two per CU (what the LDS footprint admits), a multiple of 8
    static std::atomic<int> resident{0};
    int r = resident.load(std::memory_order_relaxed);
    if (r == 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 4)
        cus = 256;
      r = (2 * cus) / 8 * 8;
      resident.store(r, std::memory_order_relaxed);
    }
    if (grid > (unsigned)r) grid = (unsigned)r;
  }
  // profiling builds: FUSED_LDS_PAD extra bytes of dynamic LDS lower the number of co-resident workgroups per CU
  hipLaunchKernelGGL((edge_layer_fused_kernel<T, ABL, NW, L0, GNP, TAIL, OPT, NOTB>), dim3(grid), dim3(64 * WV),
                     fused::Geo<NW>::LDS_TOTAL + FUSED_LDS_PAD, stream,
                     e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o, b_o, b_out,
                     time_on_edge, part, direct, FUSED_DBG, l0_table, l0_x, l0_perm, gn_tile, scales, etmax_in, etmax_out, FUSED_START_DELAY);
  return hipGetLastError();
}

// production geometry, no ablation.  Profiling builds (-DDIFUSCO_PROFILING, libdifusco_hip_prof.so) also hold the A/B
// variants of the scheduling options, selected by g_fused_opt; the production library has the production set only.
template <typename T, bool L0, bool GNP, int TAIL, bool NOTB = false, typename... A>
hipError_t launch_fused_opt(A... args) {
#ifndef DIFUSCO_PROFILING
  return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT, NOTB>(args...);
#else
  switch (g_fused_opt) {      // (the variants without bits 6 / 11 - serial sums, scalar element-wise code - and bits 13, 16, 18 were removed in round 5)
    case FUSED_OPT | 4096: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT | 4096, NOTB>(args...);    // 155507 (A/B: production + persistent workgroups)
    case 3955: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, 3955, NOTB>(args...);      // (A/B: round 2's production: register gathers)
    case FUSED_OPT | 32768: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT | 32768, NOTB>(args...);  // 184179 (A/B: ... + two gather units, counted waits)
    case 20339: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, 20339, NOTB>(args...);    // (A/B: round 3's production: no neighbour-sum fast path)
    case 150899: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, 150899, NOTB>(args...);  // (A/B: production without the raised issue priority)
    case 151411: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, 151411, NOTB>(args...);  // (A/B: rounds 2-4's production: + alternating MFMA chains, bit 1)
    case 151377: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, 151377, NOTB>(args...);  // (A/B: production without the two-stage cover of the e stream, bit 5)
    case FUSED_OPT | 2097152: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT | 2097152, NOTB>(args...);  // 2248561 (A/B: no scheduling fences around the MFMA triples)
    case FUSED_OPT | 1048576: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT | 1048576, NOTB>(args...);  // 1199985 (A/B: GEMM 2 block-major inside a stage)
    default: return launch_fused_t<T, 0, FUSED_NW, L0, GNP, TAIL, FUSED_OPT, NOTB>(args...);
  }
#endif
}

// one entry point per element type / purpose, each defined in its own translation unit.
// kind: 0 middle layer, 1 first layer from the 2-row table (L0), 2 last layer of a TSP step (GNP, TAIL 1),
//       3 last layer of a MIS step (TAIL 2);  + 4: the register-gather instantiation (n_nodes >= 2^20);
//       + 8 (kinds 0, 1, 3): aggregation = "max"
#define FUSED_KIND_PARAMS                                                                                              \
  float *e, const float *node4, const int *row, const int *col, int n_edges, const unsigned short *c_planes,          \
      const unsigned short *o_planes, long long plane_stride, const float *b_c, const float *g_e, const float *b_e,   \
      const float *tbias, const float *g_o, const float *b_o, const float *b_out, int time_on_edge, float *part,      \
      float *direct, hipStream_t stream, const float *l0_table, const float *l0_x, const int *l0_perm, float *gn_tile,  \
      const float *scales, const float *etmax_in, float *etmax_out
#define FUSED_KIND_ARGS                                                                                                \
  e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o, b_o, b_out, time_on_edge, \
      part, direct, stream, l0_table, l0_x, l0_perm, gn_tile, scales, etmax_in, etmax_out
hipError_t launch_fused_fp16(int kind, FUSED_KIND_PARAMS);
hipError_t launch_fused_bf16(int kind, FUSED_KIND_PARAMS);
hipError_t launch_fused_ablation(int mask, FUSED_KIND_PARAMS);      // profiling-only variants of the fp16 middle layer

// kinds 4-7 = kinds 0-3 with the neighbour-table rows gathered into REGISTERS by 64-bit addresses (round 2's option set):
// the full-line gathers address node rows by 32-bit byte offsets (4 KB per row), which wrap at n_nodes = 2^20.  The step
// driver (api.hip) picks these for such calls; results are bit-identical to the full-line kernels.
template <typename T>
hipError_t launch_fused_kind(int kind, FUSED_KIND_PARAMS) {
  switch (kind) {
    // (a layer without a time bias on e - MIS, gnn_encoder.py:447 - takes the NOTB instantiation: no bias reads, no adds)
    case 0: return time_on_edge ? launch_fused_opt<T, false, false, 0>(FUSED_KIND_ARGS) : launch_fused_opt<T, false, false, 0, true>(FUSED_KIND_ARGS);
    case 1: return time_on_edge ? launch_fused_opt<T, true, false, 0>(FUSED_KIND_ARGS) : launch_fused_opt<T, true, false, 0, true>(FUSED_KIND_ARGS);
    case 2: return launch_fused_opt<T, false, true, 1>(FUSED_KIND_ARGS);
    case 3: return launch_fused_opt<T, false, false, 2>(FUSED_KIND_ARGS);
    case 4: return launch_fused_t<T, 0, FUSED_NW, false, false, 0, FUSED_OPT_R2>(FUSED_KIND_ARGS);
    case 5: return launch_fused_t<T, 0, FUSED_NW, true, false, 0, FUSED_OPT_R2>(FUSED_KIND_ARGS);
    case 6: return launch_fused_t<T, 0, FUSED_NW, false, true, 1, FUSED_OPT_R2>(FUSED_KIND_ARGS);
    case 7: return launch_fused_t<T, 0, FUSED_NW, false, false, 2, FUSED_OPT_R2>(FUSED_KIND_ARGS);
    // aggregation = "max" (OPT bit 19); the last layer of a TSP step has no neighbour aggregation: kind 2 serves it
    case 8: return launch_fused_t<T, 0, FUSED_NW, false, false, 0, FUSED_OPT | 524288>(FUSED_KIND_ARGS);
    case 9: return launch_fused_t<T, 0, FUSED_NW, true, false, 0, FUSED_OPT | 524288>(FUSED_KIND_ARGS);
    case 11: return launch_fused_t<T, 0, FUSED_NW, false, false, 2, FUSED_OPT | 524288>(FUSED_KIND_ARGS);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace difusco
