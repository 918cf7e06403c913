// Internal launcher declarations (implemented in linear.hip / graph_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace difusco {

hipError_t linear_rows(const float* x, const float* w, const float* bias, const float* residual, float* y,
                       long long m, int k, int n_out, long long ldy, hipStream_t stream);

hipError_t linear_rows_split(const float* x, const unsigned short* wp, long long plane_stride, int mode,
                             const float* bias, const float* residual, float* y, long long m, int k, int n_out,
                             long long ldy, hipStream_t stream);

hipError_t launch_edge_layer_fused(int mode, float* e, const float* node4, const int* row, const int* col, int n_edges,
                                   const unsigned short* c_planes, const unsigned short* o_planes,
                                   long long plane_stride, const float* b_c, const float* g_e, const float* b_e,
                                   const float* tbias, const float* g_o, const float* b_o, const float* b_out,
                                   int time_on_edge, float* part, float* direct, hipStream_t stream);
extern int g_fused_ablate;
extern unsigned long long* g_fused_dbg;
hipError_t launch_node_finalize(int n_nodes, int n_edges, const int* rowptr, const float* node4, const float* part,
                                const float* direct, float* h, const float* nh_w, const float* nh_b,
                                const float* tbias, int time_on_edge, hipStream_t stream);

hipError_t launch_time_bias(float t, int H, int n_layers, const float* freqs, const float* w0, const float* b0,
                            const float* w2, const float* b2, const float* wl_base, long long layer_stride,
                            long long wl_w_off, long long wl_b_off, float* tbias, hipStream_t stream);
hipError_t launch_pos_embed(const float* points, const float* dimt, int n_nodes, int H, float* out, hipStream_t stream);
hipError_t launch_scalar_embed(const float* x, const int* perm, const float* dimt, long long rows, int H, float* out,
                               hipStream_t stream);
hipError_t launch_table_rows(const float* x, const int* perm, const float* table, long long rows, int H, float* out,
                             hipStream_t stream);
hipError_t launch_edge_gate_aggregate(int H, int n_nodes, const int* rowptr, const int* col, const float* node4,
                                      float* ce_act, float* h, const float* nh_w, const float* nh_b, const float* ne_w,
                                      const float* ne_b, const float* ol_w, const float* ol_b, const float* tbias,
                                      int time_on_edge, hipStream_t stream);
int gn_blocks_for(long long rows);
hipError_t launch_head(int H, int C, const float* feat, const int* seg_ptr, int n_segments, long long total_rows,
                       int nblk, double* partial, float* stats, const float* gn_w, const float* gn_b,
                       const float* conv_w, const float* conv_b, const int* perm, const float* xt, const float* post,
                       int rand_mode, const float* rand, unsigned long long seed, unsigned long long offset,
                       float* xt_out, float* pred_out, float* prob_out, hipStream_t stream);
hipError_t launch_categorical_posterior(const float* logits, const float* xt, const float* post, int rand_mode,
                                        const float* rand, unsigned long long seed, unsigned long long offset,
                                        float* xt_out, float* prob_out, long long n, hipStream_t stream);
hipError_t launch_gaussian_posterior(const float* pred, const float* xt, const float* post, int rand_mode,
                                     const float* rand, unsigned long long seed, unsigned long long offset, float* xt_out,
                                     long long n, hipStream_t stream);

}  // namespace difusco
