// Fused edge pass of one DIFUSCO GNN layer for gfx950 (H = 256): launchers, the node update that follows the edge pass,
// and the fp16 production instantiations of the kernel template in edge_layer_kernel.h.
#include "edge_layer_kernel.h"

namespace difusco {

// ------------------------------------------------------------------------------------------------
// node update after the fused edge pass:  h_i += ReLU(LN_h(Uh_i + sum_j gate*Vh_j)) (+ t, MIS)
// (gnn_encoder.py:115,123,134,447-448).  One wavefront per node; the neighbour sum is assembled from the
// per-tile pieces written by edge_layer_fused_kernel, in tile order.  agg_mode 1 (aggregation = "mean",
// gnn_encoder.py:170-171,184-185): the sum is divided by the number of edges of the row (empty row: 0); agg_mode 2 ("max",
// :172-173,187-188): the pieces are maxima and are combined by maximum (empty row: 0).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_finalize_kernel(int n_nodes, int n_edges, const int* __restrict__ rowptr,
                                                            const float* __restrict__ node4,
                                                            const float* __restrict__ part,
                                                            const float* __restrict__ direct, float* h,
                                                            const float* __restrict__ nh_w,
                                                            const float* __restrict__ nh_b,
                                                            const float* __restrict__ tbias, int time_on_edge,
                                                            float* __restrict__ row_scale, const float* h_in,
                                                            int agg_mode) {
  constexpr int H = 256;
  const int lane = threadIdx.x & 63;
  const int f = lane * 4;
  const int i = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (i >= n_nodes) return;
  const int a = rowptr[i], b = rowptr[i + 1];
  const bool agg_mean = agg_mode == 1, agg_max = agg_mode == 2 && b > a;      // (wave uniform)
  v4f agg = {0.0f, 0.0f, 0.0f, 0.0f};
  if (agg_max) agg = v4f{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  if (b > a) {
    const int t0 = a >> 5, t1 = (b - 1) >> 5;
    for (int t = t0; t <= t1; ++t) {
      const int first = 32 * t;
      int last = first + 31;
      last = last < n_edges ? last : n_edges - 1;
      const float* src;
      if (a <= first) src = part + ((long long)t * 2 + 0) * H;          // owns the tile's first edge
      else if (b > last) src = part + ((long long)t * 2 + 1) * H;       // owns the tile's last edge
      else src = direct + (long long)i * H;                              // strictly inside the tile
      const v4f piece = *reinterpret_cast<const v4f*>(src + f);
      if (agg_max) {      // (the pieces are maxima: kinds 8, 9, 11 of the fused kernel)
#pragma unroll
        for (int q = 0; q < 4; ++q) agg[q] = __builtin_fmaxf(agg[q], piece[q]);
      } else {
        agg += piece;
      }
    }
    if (agg_mean) {      // (wave uniform) true divisions: the value of sum / count
      const float cnt = (float)(b - a);
#pragma unroll
      for (int q = 0; q < 4; ++q) agg[q] = agg[q] / cnt;
    }
  }
  const v4f uh = *reinterpret_cast<const v4f*>(node4 + (long long)i * 4 * H + f);
  v4f x = uh + agg;
  constexpr float inv_h = 1.0f / 256.0f;
  const float mean = wave_sum(x[0] + x[1] + x[2] + x[3]) * inv_h;
  v4f d = x - mean;
  const float rstd = 1.0f / sqrtf(wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * inv_h + 1e-5f);
  const v4f gw = *reinterpret_cast<const v4f*>(nh_w + f), gb = *reinterpret_cast<const v4f*>(nh_b + f);
  const v4f tb = *reinterpret_cast<const v4f*>(tbias + f);
  float* hp = h + (long long)i * H + f;
  v4f hv = *reinterpret_cast<const v4f*>(h_in + (long long)i * H + f);      // (h_in == h: in place)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float y = d[q] * rstd * gw[q] + gb[q];
    y = y > 0.0f ? y : 0.0f;
    if (!time_on_edge) y += tb[q];
    hv[q] += y;
  }
  *reinterpret_cast<v4f*>(hp) = hv;
  if (row_scale != nullptr) {      // power-of-two operand scale of this row for the next node-row linear (fp16 planes)
    float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(hv[0]), __builtin_fabsf(hv[1])),
                              __builtin_fmaxf(__builtin_fabsf(hv[2]), __builtin_fabsf(hv[3])));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
    float inv;
    const float sc = pow2_scale_for(m, inv);
    if (lane == 0) row_scale[i] = sc;
  }
}

#ifdef DIFUSCO_PROFILING
// process-wide knobs of the PROFILING library (libdifusco_hip_prof.so, difusco_debug_set); the production library has none
int g_fused_ablate = 0;   // ablation mask (difusco_debug_set key 0): wrong results by construction, timing only
int g_fused_lds_pad = 0;  // extra dynamic LDS bytes: occupancy probe (key 6)
int g_fused_start_delay = 0;   // cycles by which the second-slot workgroups of the first generation start late (key 9)
int g_fused_opt = FUSED_OPT;   // key 7: A/B variants of the scheduling options
unsigned long long* g_fused_dbg = nullptr;   // device buffer for phase timestamps, [n_tiles][16]
#endif

hipError_t launch_fused_fp16(int kind, FUSED_KIND_PARAMS) { return launch_fused_kind<FFp16>(kind, FUSED_KIND_ARGS); }

// variant bit 0: the register-gather instantiation of the same kind (kind + 4; calls with n_nodes >= 2^20);
// variant bit 1: aggregation = "max" (kind + 8 for the kinds that aggregate; not combined with bit 0)
static hipError_t launch_by_mode(int mode, int kind, int variant, FUSED_KIND_PARAMS) {
  if (n_edges <= 0) return hipSuccess;
  if (variant & 2) {
    if (variant & 1) return hipErrorInvalidValue;
    if (kind != 2) kind += 8;
  } else if (variant & 1) {
    kind += 4;
  }
  if (mode == 1) return launch_fused_bf16(kind, FUSED_KIND_ARGS);     // DIFUSCO_PREC_BF16X3
  if (mode == 3) return launch_fused_fp16(kind, FUSED_KIND_ARGS);     // DIFUSCO_PREC_FP16X3
  return hipErrorInvalidValue;
}

// mode: 1 = bf16 planes, 3 = fp16 planes (DIFUSCO_PREC_BF16X3 / DIFUSCO_PREC_FP16X3)
hipError_t launch_edge_layer_fused(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                   const unsigned short* c_planes, const unsigned short* o_planes,
                                   long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                   const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                   int time_on_edge, float* part, float* direct, const float* scales,
                                   const float* etmax_in, float* etmax_out, hipStream_t stream, int reg_gather) {
  const float *l0_table = nullptr, *l0_x = nullptr;
  const int* l0_perm = nullptr;
  float* gn_tile = nullptr;
#ifdef DIFUSCO_PROFILING
  if (mode == 3 && g_fused_ablate != 0) {      // profiling-only variants exist for the fp16 middle layer
    if (n_edges <= 0) return hipSuccess;
    return launch_fused_ablation(g_fused_ablate, FUSED_KIND_ARGS);
  }
#endif
  return launch_by_mode(mode, 0, reg_gather, FUSED_KIND_ARGS);
}

// Last layer of a step.  tail 1 (TSP: the head normalises e): the per-tile GroupNorm partial sums
// gn_tile[ceil(n_edges / 32)][32][2] of the new edge state are emitted and the node update is skipped (no
// node_finalize after it).  tail 2 (MIS: the head reads h): the kernel stops after the neighbour sum, e is not updated.
hipError_t launch_edge_layer_fused_tail(int mode, int tail, float* e, const float* node4, const int* row, const int* col,
                                        int n_edges, const unsigned short* c_planes, const unsigned short* o_planes,
                                        long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                        const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                        int time_on_edge, float* part, float* direct, float* gn_tile,
                                        const float* scales, const float* etmax_in, hipStream_t stream, int reg_gather) {
  const float *l0_table = nullptr, *l0_x = nullptr;
  const int* l0_perm = nullptr;
  float* etmax_out = nullptr;
  if (tail != 1 && tail != 2) return hipErrorInvalidValue;
  return launch_by_mode(mode, tail == 1 ? 2 : 3, reg_gather, FUSED_KIND_ARGS);
}

// First layer of a step whose edge input is a table lookup (see the L0 notes in the kernel): same as
// launch_edge_layer_fused, but e is only written.  table: [4][256] floats (rows 0, 1 the input rows, rows 2, 3 C applied
// to them), x: per caller-edge values (null = row 0), perm: CSR slot -> caller edge id (null = identity).
hipError_t launch_edge_layer_fused_l0(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                      const unsigned short* c_planes, const unsigned short* o_planes,
                                      long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                      const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                      int time_on_edge, float* part, float* direct, const float* table, const float* x,
                                      const int* perm, const float* scales, float* etmax_out, hipStream_t stream,
                                      int reg_gather) {
  const float *l0_table = table, *l0_x = x;
  const int* l0_perm = perm;
  float* gn_tile = nullptr;
  const float* etmax_in = nullptr;
  return launch_by_mode(mode, 1, reg_gather, FUSED_KIND_ARGS);
}

// The reference's node rows -> the fused kernel's log2(e) domain (only the stand-alone layer entry difusco_edge_layer_fused needs
// this pass: the step driver's node linear writes the rows in that form).  One thread per float4.
__global__ __launch_bounds__(256) void fuse_node_tables_kernel(const float* __restrict__ node4, const float* __restrict__ b_c,
                                                               long long n_vec, float* __restrict__ out) {
  constexpr float kLog2e = 1.4426950408889634f;
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vec) return;
  const int c = (int)(v & 255) * 4;      // column inside the 1024-float row
  v4f x = *reinterpret_cast<const v4f*>(node4 + v * 4);
  if (c >= 512) {
    if (c < 768) x += *reinterpret_cast<const v4f*>(b_c + (c - 512));
    x = x * kLog2e;
  }
  *reinterpret_cast<v4f*>(out + v * 4) = x;
}

hipError_t launch_fuse_node_tables(const float* node4, const float* b_c, int n_nodes, float* out, hipStream_t stream) {
  if (n_nodes <= 0) return hipSuccess;
  const long long n_vec = (long long)n_nodes * 256;
  hipLaunchKernelGGL(fuse_node_tables_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, stream, node4, b_c, n_vec, out);
  return hipGetLastError();
}

hipError_t launch_node_finalize(int n_nodes, int n_edges, const int* rowptr, const float* node4, const float* part,
                                const float* direct, float* h, const float* nh_w, const float* nh_b,
                                const float* tbias, int time_on_edge, float* row_scale, hipStream_t stream,
                                const float* h_in, int agg_mode) {
  if (n_nodes <= 0) return hipSuccess;
  if (agg_mode < 0 || agg_mode > 2) return hipErrorInvalidValue;
  hipLaunchKernelGGL(node_finalize_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, stream, n_nodes, n_edges,
                     rowptr, node4, part, direct, h, nh_w, nh_b, tbias, time_on_edge, row_scale, h_in ? h_in : h, agg_mode);
  return hipGetLastError();
}

}  // namespace difusco
