// Internal launcher declarations (implemented in linear.hip / graph_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace difusco {

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `devices` is the per-kernel bit mask of the
// devices already configured.  Thread safe (setting the attribute twice is harmless).
inline hipError_t ensure_max_dynamic_lds(std::atomic<unsigned long long>& devices, const void* fn, int bytes) {
  int dev = 0;
  hipError_t er = hipGetDevice(&dev);
  if (er != hipSuccess) return er;
  const unsigned long long bit = 1ull << (dev & 63);
  if (devices.load(std::memory_order_acquire) & bit) return hipSuccess;
  er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (er != hipSuccess) return er;
  devices.fetch_or(bit, std::memory_order_release);
  return hipSuccess;
}

// sets the message returned by difusco_last_error() and returns `code` (api.hip)
int set_error(int code, const char* fmt, ...);

// Tiled ("MFMA native") layout of an [E, 256] fp32 edge-feature matrix, used by the fused path.
// Tile = 32 consecutive edges = 8192 floats laid out [slab ks = f/16 (16)][i = (f/8)%2][lane = ((f/4)%2)*32 + s%32 (64)][q = f%4]:
// the float4 that lane (s%32, hh) of a wavefront feeds to / receives from a 32x32x16 MFMA for slab ks is at
//   tile*8192 + ks*512 + i*256 + lane*4,
// so one wave instruction moves one contiguous KiB.  Rows are padded to a multiple of 256 edges; pad lanes hold 0.
__host__ __device__ inline long long edge_tiled_offset(long long s, int f) {
  return (s >> 5) * 8192 + (long long)(f >> 4) * 512 + ((f >> 3) & 1) * 256 + ((((f >> 2) & 1) * 32) + (s & 31)) * 4 + (f & 3);
}

hipError_t linear_rows(const float* x, const float* w, const float* bias, const float* residual, float* y,
                       long long m, int k, int n_out, long long ldy, hipStream_t stream);

// Operand scaling of the fp16 split path (all exact powers of two; defaults = no scaling, which is what the bf16
// modes use): X row r is multiplied by x_scale * row_scale[r] before the split, the planes hold W[f] / w_inv[f], and the
// accumulator is multiplied by the inverses where the bias is added.  tile_max (tiled output, n_out == 256): max |Y| per
// 32-row tile - the per-tile e-stream scale the fused edge kernel reads.
struct SplitScale {
  const float* row_scale = nullptr;
  float x_scale = 1.0f;
  const float* w_inv = nullptr;
  float* tile_max = nullptr;
};

// tiled_out != 0 (k == n_out == 256 only): Y is written in the tiled layout above instead of row major
hipError_t linear_rows_split(const float* x, const unsigned short* wp, long long plane_stride, int mode,
                             const float* bias, const float* residual, float* y, long long m, int k, int n_out,
                             long long ldy, hipStream_t stream, int tiled_out = 0, SplitScale sc = SplitScale());

// the same product for k = 256, n_out % 128 == 0, modes 1 / 3, row-major y with ldy = n_out, no residual (node_linear.hip):
// rows register resident, weight planes streamed through LDS.  linear_rows_split routes the node-row shape here.
hipError_t node_linear(const float* x, const float* row_scale, long long m, const unsigned short* planes, long long plane_stride,
                       int mode, int n_out, const float* w_inv, const float* bias, float* y, hipStream_t stream);

// Y = ScalarEmbeddingSine(x[perm]) W^T + b, the [m,256] embedding generated inside the kernel (split planes, modes 1 / 3)
hipError_t linear_scalar_embed_split(const float* x, const int* perm, const float* dimt, const unsigned short* wp,
                                     long long plane_stride, int mode, const float* bias, float* y, long long m,
                                     hipStream_t stream, int tiled_out, const float* w_inv, float* tile_max);

// the same product for the FUSED path (edge_embed.hip): e0 written in the tiled edge layout with its per-tile max |e|; the dataflow of
// the fused edge kernel's GEMM 1 (weights streamed through LDS by LDS-DMA, the generated rows as the register-resident MFMA operand)
hipError_t launch_edge_embed_tiled(const float* x, const int* perm, const float* dimt, const unsigned short* planes, long long plane_stride,
                                   int mode, const float* w_inv, const float* bias, float* e, long long n_edges, float* tile_max,
                                   hipStream_t stream, const float* gen_table = nullptr, int* tile_flag = nullptr);
// The generated-input table (round 6, edge_embed.hip): e0(x) = edge_embed(ScalarEmbeddingSine(x)) sampled at x_r = kGenXMin + (r - 3) kGenH,
// r < kGenRows; cell c = floor((x - kGenXMin) / kGenH) in [0, kGenIntervals) interpolates rows c .. c + 7 (degree-7 Lagrange).
constexpr float kGenXMin = -8.0f, kGenH = 0.25f, kGenInvH = 4.0f;
constexpr int kGenIntervals = 64, kGenRows = kGenIntervals + 7;
hipError_t launch_gen_table_grid(float* x, hipStream_t stream);

// scale[r] = power-of-two operand scale of row r of x[m][k] (fp16 split path)
hipError_t launch_row_pow2_scale(const float* x, long long m, int k, float* scale, hipStream_t stream);
// *count += number of inf / nan values in x[0..n)  (DIFUSCO_FLAG_CHECK_FINITE)
hipError_t launch_count_nonfinite(const float* x, long long n, unsigned* count, hipStream_t stream);
// tile_max[t] = max |e| over the 32-edge tile t of a TILED [rows_padded, 256] buffer
hipError_t launch_tile_absmax_tiled(const float* e, long long n_tiles, float* tile_max, hipStream_t stream);

hipError_t launch_edge_layer_fused(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                   const unsigned short* c_planes, const unsigned short* o_planes,
                                   long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                   const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                   int time_on_edge, float* part, float* direct, const float* scales,
                                   const float* etmax_in, float* etmax_out, hipStream_t stream, int reg_gather = 0);
hipError_t launch_edge_layer_fused_l0(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                      const unsigned short* c_planes, const unsigned short* o_planes,
                                      long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                      const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                      int time_on_edge, float* part, float* direct, const float* table, const float* x,
                                      const int* perm, const float* scales, float* etmax_out, hipStream_t stream,
                                      int reg_gather = 0);
hipError_t launch_edge_layer_fused_tail(int mode, int tail, float* e, const float* node4, const int* row, const int* col,
                                        int n_edges, const unsigned short* c_planes, const unsigned short* o_planes,
                                        long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                        const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                        int time_on_edge, float* part, float* direct, float* gn_tile,
                                        const float* scales, const float* etmax_in, hipStream_t stream, int reg_gather = 0);
// reg_gather != 0: the variant of the kernel that gathers neighbour-table rows into registers by 64-bit addresses - for calls
// with n_nodes >= 2^20, where the 32-bit byte offsets of the full-line (LDS-DMA) gathers would wrap (4 KB per node row)
#ifdef DIFUSCO_PROFILING
extern int g_fused_ablate;
extern int g_fused_lds_pad;
extern int g_fused_start_delay;
extern int g_fused_opt;
extern int g_node_linear_depth;
extern int g_node_linear_ablate;
extern unsigned long long* g_fused_dbg;
#endif
// h_in: the node state the update is added to (nullptr = h itself, in place; layer 0 of a step with prepared state reads the
// prepared h0 and writes the workspace h)
// node4 [n, 4*256] (U | V | A | B rows as the reference defines them) -> out: U | V copied, (A + b_c) log2(e), B log2(e): the form in
// which the fused edge kernel reads its neighbour tables (difusco_hip.h, ABI 11 note)
hipError_t launch_fuse_node_tables(const float* node4, const float* b_c, int n_nodes, float* out, hipStream_t stream);
hipError_t launch_node_finalize(int n_nodes, int n_edges, const int* rowptr, const float* node4, const float* part,
                                const float* direct, float* h, const float* nh_w, const float* nh_b,
                                const float* tbias, int time_on_edge, float* row_scale, hipStream_t stream,
                                const float* h_in = nullptr, int agg_mode = 0);
// agg_mode (DIFUSCO_AGG_*): 0 sum, 1 mean (sum / row length), 2 max (unfused kernel only) - gnn_encoder.py:170-191

// tbias[i][l][:] = time_layer_l(time_embed(timestep_embedding(t_host[i]))) for n_t diffusion times (HOST array)
hipError_t launch_time_bias(const float* t_host, int n_t, int H, int n_layers, const float* freqs, const float* w0,
                            const float* b0, const float* w2, const float* b2, const float* wl_base, long long layer_stride,
                            long long wl_w_off, long long wl_b_off, float* tbias, hipStream_t stream);
hipError_t launch_pos_embed(const float* points, const float* dimt, int n_nodes, int H, float* out, hipStream_t stream);
hipError_t launch_scalar_embed(const float* x, const int* perm, const float* dimt, long long rows, int H, float* out,
                               hipStream_t stream);
// y[2][H] = x[2][H] W^T + b for the two-row layer-0 tables (tiny: one wavefront per output feature)
hipError_t launch_two_rows_linear(int H, const float* x, const float* w, const float* bias, float* y, hipStream_t stream);
hipError_t launch_table_rows(const float* x, const int* perm, const float* table, long long rows, int H, float* out,
                             hipStream_t stream);
hipError_t launch_table_rows_tiled(const float* x, const int* perm, const float* table, long long rows, float* out,
                                   hipStream_t stream);
hipError_t launch_head_tiled(int C, const float* feat, long long rows, int nblk, double* partial, float* stats,
                             const float* gn_w, const float* gn_b, const float* conv_w, const float* conv_b,
                             const int* perm, const float* xt, const float* post, int rand_mode, const float* rand,
                             unsigned long long seed, unsigned long long offset, float* xt_out, float* pred_out,
                             float* prob_out, hipStream_t stream, const float* gn_tile = nullptr, int gn_phase = 0,
                             double* gn_sums = nullptr, const int* seg_ptr = nullptr, int n_segments = 1);
hipError_t launch_edge_gate_aggregate(int H, int n_nodes, const int* rowptr, const int* col, const float* node4,
                                      float* ce_act, float* h, const float* nh_w, const float* nh_b, const float* ne_w,
                                      const float* ne_b, const float* ol_w, const float* ol_b, const float* tbias,
                                      int time_on_edge, hipStream_t stream, int agg_mode = 0);
int gn_blocks_for(long long rows);
hipError_t launch_head(int H, int C, const float* feat, const int* seg_ptr, int n_segments, long long total_rows,
                       int nblk, double* partial, float* stats, const float* gn_w, const float* gn_b,
                       const float* conv_w, const float* conv_b, const int* perm, const float* xt, const float* post,
                       int rand_mode, const float* rand, unsigned long long seed, unsigned long long offset,
                       float* xt_out, float* pred_out, float* prob_out, hipStream_t stream, int gn_phase = 0,
                       double* gn_sums = nullptr);
hipError_t launch_categorical_posterior(const float* logits, const float* xt, const float* post, int rand_mode,
                                        const float* rand, unsigned long long seed, unsigned long long offset,
                                        float* xt_out, float* prob_out, long long n, hipStream_t stream);
hipError_t launch_gaussian_posterior(const float* pred, const float* xt, const float* post, int rand_mode,
                                     const float* rand, unsigned long long seed, unsigned long long offset, float* xt_out,
                                     long long n, hipStream_t stream);

}  // namespace difusco
