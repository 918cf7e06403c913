/*
 * difusco_hip.h - C ABI of the MI355X-native (gfx950) DIFUSCO denoise-step library.
 *
 * Drop-in boundary.  The reference has no FFI: its boundary is two Python methods,
 *   TSPModel.categorical_denoise_step / gaussian_denoise_step   (difusco/pl_tsp_model.py:122-151)
 *   MISModel.categorical_denoise_step / gaussian_denoise_step   (difusco/pl_mis_model.py:118-140)
 * which call GNNEncoder.forward (difusco/models/gnn_encoder.py:452-462) and
 * COMetaModel.categorical_posterior / gaussian_posterior (difusco/pl_meta_model.py:102-175).
 * This header declares what a binding for that path needs: plain pointers and sizes, no torch types.
 * All device pointers are HIP device memory of the current device; `stream` is a hipStream_t.
 * Every function returns 0 on success, a negative DIFUSCO_E* code otherwise; difusco_last_error()
 * gives the message of the calling thread's last failure.
 *
 * The binding a reference maintainer would add (ctypes) is shown in INTEGRATION.md; the in-tree
 * Python host side is difusco_amd/ (same method names and argument meaning as the reference).
 */
#ifndef DIFUSCO_HIP_H
#define DIFUSCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIFUSCO_ABI_VERSION 12

enum {
  DIFUSCO_OK = 0,
  DIFUSCO_EINVAL = -1,      /* bad argument (shape, null pointer, unsupported hidden size ...) */
  DIFUSCO_EWORKSPACE = -2,  /* workspace too small */
  DIFUSCO_EHIP = -3,        /* a HIP runtime call failed */
  DIFUSCO_EUNSUPPORTED = -4, /* a combination the library has no kernel for */
  DIFUSCO_ENONFINITE = -5    /* DIFUSCO_FLAG_CHECK_FINITE: the step produced inf / nan */
};

enum { DIFUSCO_TASK_TSP = 0, DIFUSCO_TASK_MIS = 1 };            /* edge features | node features only */
enum { DIFUSCO_CATEGORICAL = 0, DIFUSCO_GAUSSIAN = 1 };
/* neighbourhood aggregation of the GNN layers (--aggregation, train.py:52; gnn_encoder.py:170-191): every published run uses sum.
 * MAX combines messages with fmaxf, which returns the non-NaN operand: a NaN message is DROPPED where torch.max / segment_csr(max)
 * would propagate it into h.  A non-finite state still surfaces through e (and DIFUSCO_FLAG_CHECK_FINITE sees it there). */
enum { DIFUSCO_AGG_SUM = 0, DIFUSCO_AGG_MEAN = 1, DIFUSCO_AGG_MAX = 2 };
enum {
  DIFUSCO_PREC_FP32 = 0,   /* E-row linears on v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fma chain) */
  DIFUSCO_PREC_BF16X3 = 1, /* 2 bf16 planes, 3 products: ~2^-17 relative per product                    */
  DIFUSCO_PREC_BF16X6 = 2, /* 3 bf16 planes, 6 products: all 24 significand bits, fp32-class accuracy    */
  DIFUSCO_PREC_FP16X3 = 3  /* 2 fp16 planes, 3 products, operands pre-scaled by exact powers of two into the upper
                              binades of fp16 (weights per matrix / per row on the host, activations per row or per
                              32-edge tile on the device): 22 significand bits for elements within 2^-17 of the
                              largest element of their scaling group, an absolute floor of 2^-39 of that largest
                              element below; any finite fp32 operand scale (no |x| < 65504 restriction)            */
};
enum {
  DIFUSCO_RAND_NONE = 0,     /* no draw: categorical final step (target_t == 0) or DDIM */
  DIFUSCO_RAND_INJECTED = 1, /* caller supplies uniforms (categorical) / normals (gaussian DDPM) */
  DIFUSCO_RAND_PHILOX = 2    /* on-device Philox4x32-10 keyed by (seed, offset, element) */
};

int difusco_abi_version(void);
const char* difusco_last_error(void);

/* ---- weights ---------------------------------------------------------------------------------
 * The packed fp32 blob holds every tensor of the reference GNNEncoder state_dict
 * (gnn_encoder.py:303-347; key set in SURVEY.md section 5) plus three host-computed constant
 * tables (timestep frequencies nn.py:114-116, the two `dim_t` vectors gnn_encoder.py:215,243).
 * difusco_weights_layout() is the single source of truth for the offsets (in floats). */
enum {
  DIFUSCO_W_NODE_EMBED_W = 0, DIFUSCO_W_NODE_EMBED_B,
  DIFUSCO_W_EDGE_EMBED_W, DIFUSCO_W_EDGE_EMBED_B,
  DIFUSCO_W_TIME0_W, DIFUSCO_W_TIME0_B, DIFUSCO_W_TIME2_W, DIFUSCO_W_TIME2_B,
  DIFUSCO_W_OUT_GN_W, DIFUSCO_W_OUT_GN_B, DIFUSCO_W_OUT_CONV_W, DIFUSCO_W_OUT_CONV_B,
  DIFUSCO_W_TIME_FREQS,  /* [H/2]  exp(-ln(1e4) k/(H/2))                      */
  DIFUSCO_W_DIMT_POS,    /* [H/2]  1e4^(2(k/2)/(H/2))  PositionEmbeddingSine   */
  DIFUSCO_W_DIMT_SCALAR, /* [H]    1e4^(2(k/2)/H)      ScalarEmbeddingSine(1D).  REQUIRED: entries 2j and 2j+1 are EQUAL (the
                            reference's table, gnn_encoder.py:243): the generated-input kernels (edge_embed.hip, the GEN path of
                            linear_split.hip) take ONE sincos of x / dim_t[2j] for the (sin, cos) feature pair (2j, 2j+1); a blob
                            with another table diverges from scalar_embed_kernel.  pack_state_dict() asserts it. */
  DIFUSCO_W_EDGE_EMBED_PLANES, /* bf16 split planes of edge_embed.weight, see below */
  DIFUSCO_W_GLOBAL_COUNT
};
enum {                       /* per layer l, index = GLOBAL_COUNT + l*LAYER_COUNT + id */
  DIFUSCO_WL_NODE4_W = 0,    /* [4H,H] rows: U | V | A | B  (gnn_encoder.py:52-55)  */
  DIFUSCO_WL_NODE4_B,        /* [4H]                                                */
  DIFUSCO_WL_C_W, DIFUSCO_WL_C_B,
  DIFUSCO_WL_NORM_H_W, DIFUSCO_WL_NORM_H_B, DIFUSCO_WL_NORM_E_W, DIFUSCO_WL_NORM_E_B,
  DIFUSCO_WL_TIME_W, DIFUSCO_WL_TIME_B,      /* time_embed_layers[l].1 : [H,H/2],[H] */
  DIFUSCO_WL_OUT_LN_W, DIFUSCO_WL_OUT_LN_B,  /* per_layer_out[l].0                   */
  DIFUSCO_WL_OUT_W, DIFUSCO_WL_OUT_B,        /* per_layer_out[l].2 : [H,H],[H]       */
  DIFUSCO_WL_C_PLANES, DIFUSCO_WL_OUT_PLANES, /* split planes of C / per_layer_out[l].2 */
  DIFUSCO_WL_NODE4_PLANES,                    /* split planes of the [4H,H] node linear (U|V|A|B) */
  DIFUSCO_WL_FUSED_SCALES,                    /* 8 floats: operand scales of the fused edge kernel, see below */
  DIFUSCO_WL_NODE4_FUSED_B,                   /* [4H]   bias of the node linear as the FUSED edge kernel wants its rows, see below (ABI 11) */
  DIFUSCO_WL_NODE4_FUSED_S,                   /* [2][4H] column scales of the node linear for the fused path: fp16 planes | bf16 planes   */
  DIFUSCO_WL_COUNT
};
/* "*_PLANES" entries: the [H,H] weight w decomposed on the host into five 16-bit planes
 *   bf16: hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)      (round to nearest even)
 *   fp16: hi = fp16(w), lo = fp16(w - hi)
 * stored back to back in that order (plane stride H*H elements), each plane laid out
 * [H/16 slabs][H rows][16] with slab position j holding k = 16 s + {0..3, 8..11, 4..7, 12..15}[j]
 * (the order in which an MFMA accumulator feeds the next MFMA, see linear_split.hip).
 * The fp16 planes are those of the SCALED matrix w[f][:] * 2^k_f, k_f chosen on the host so that the largest scaled
 * magnitude of the scaling group lies in [2^14, 2^15) (one group per matrix for C / per_layer_out / edge_embed, one per
 * output row for the node linear); the n_out floats that follow the five planes hold 2^-k_f.  Without the scale the
 * low plane of any |w| < 2^-3 would be an fp16 subnormal with an absolute 2^-25 floor.  The bf16 planes are unscaled
 * (bf16 has the fp32 exponent range).
 * 5*n_out*k 16-bit elements + n_out floats = 2.5*n_out*k + n_out floats of blob space per matrix.  They feed the
 * split-precision MFMA paths selected by difusco_step_args.precision.
 * DIFUSCO_WL_FUSED_SCALES = {log2(e) 2^-kc, 2^-(ko+ka) / log2(e), log2(e), 2^-ka, 0, 0, 0, 0}: kc / ko the plane scales of C /
 * per_layer_out[l][2]; 2^ka the scale under which the fused kernel produces the GEMM 2 operand a log2(e), a = SiLU(LN_o(.)), from
 * the bound |a| <= max(16 max|g_o| + max|b_o|, 0.2785) (|LayerNorm| <= sqrt(H-1) < 16).  difusco_amd/weights.py computes
 * all of it (fused_scales); the e operand of GEMM 1 is scaled per 32-edge tile on the device.
 * ABI 11, "log2(e) domain": the fused edge kernel carries the gate pre-activation e' = A h[j] + B h[i] + C e + b_C as e' log2(e)
 * (sigmoid = 1 / (1 + exp2(-.)) without a multiply; LayerNorm is scale invariant).  Its neighbour-table rows node4[:, 2H:4H]
 * therefore hold (A h + b_A + b_C) log2(e) | (B h + b_B) log2(e): on the fused path the node linear is run with
 * DIFUSCO_WL_NODE4_FUSED_B = {b_U, b_V, (b_A + b_C) log2(e), b_B log2(e)} as its bias and with the per-column accumulator scales
 * DIFUSCO_WL_NODE4_FUSED_S (row 0: 2^-k_f of the fp16 planes, times log2(e) on the A | B columns; row 1, unscaled bf16 planes:
 * 1 | 1 | log2(e) | log2(e)).  The U | V columns are unchanged.  The unfused kernels read node4 as the reference defines it. */
/* Fills offsets[0 .. GLOBAL_COUNT + n_layers*WL_COUNT) (floats from blob start) and *total_floats.
 * Returns the number of entries, or a negative error. */
int difusco_weights_layout(int hidden, int n_layers, int out_channels,
                           int64_t* offsets, int max_entries, int64_t* total_floats);

/* ---- graph -----------------------------------------------------------------------------------
 * HOST helper (no GPU needed): COO int64 edge list (2 x E, reference layout
 * co_datasets/tsp_graph_dataset.py:53-62, mis_dataset.py:43-48, duplicate_edge_index
 * pl_meta_model.py:177-184) -> CSR over the centre node edge_index[0], stable in the caller's edge
 * order.  rowptr[n_nodes+1], col[E] (= edge_index[1] in CSR order), row[E], perm[E] (CSR slot ->
 * caller edge id).  *identity = 1 when the input was already row-sorted (perm[s] == s). */
int difusco_csr_from_coo_host(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes,
                              int32_t* rowptr, int32_t* col, int32_t* row, int32_t* perm,
                              int* identity);

/* ---- the denoise step -------------------------------------------------------------------------- */
typedef struct difusco_step_args {
  uint32_t struct_size;   /* = sizeof(difusco_step_args), checked */
  uint32_t abi_version;   /* = DIFUSCO_ABI_VERSION */

  /* model */
  int32_t hidden;         /* 64 | 128 | 256 (reference default 256, train.py:50) */
  int32_t n_layers;
  int32_t out_channels;   /* 2 categorical | 1 gaussian (pl_meta_model.py:28-35) */
  int32_t task;           /* DIFUSCO_TASK_* */
  const float* weights;   /* device, packed per difusco_weights_layout */

  /* graph: disjoint union of the graphs of this call, CSR over centre node (device) */
  int32_t n_nodes, n_edges;
  const int32_t* rowptr;  /* [n_nodes+1] */
  const int32_t* col;     /* [n_edges] */
  const int32_t* perm;    /* [n_edges] CSR slot -> caller edge id, or NULL (identity) */
  /* statistic segments of the head GroupNorm over output rows (edges for TSP, nodes for MIS), in
   * CSR-slot / node order.  n_segments = 1, seg_ptr = {0, rows} reproduces the reference's sparse
   * call (one statistic over ALL graphs of the call, gnn_encoder.py:400-401); one
   * segment per graph reproduces the dense path (gnn_encoder.py:380).  Device pointer; may be
   * NULL when n_segments == 1. */
  int32_t n_segments;
  const int32_t* seg_ptr; /* device, [n_segments+1] */

  /* inputs (device) */
  const float* points;    /* [n_nodes,2]  TSP only */
  const float* xt;        /* TSP: [n_edges] caller edge order;  MIS: [n_nodes] */
  float t;                /* diffusion time t (shared by the whole call, pl_tsp_model.py:124) */
  int32_t xt_is_binary;   /* 1: xt is exactly 0/1 (categorical inference): 2-row embedding table */

  /* posterior */
  int32_t diffusion;      /* DIFUSCO_CATEGORICAL | DIFUSCO_GAUSSIAN */
  /* categorical: post[0..3] = c0[xt=0], c0[xt=1], c1[xt=0], c1[xt=1] with
   *   p(x_s=1) = c0[xt]*p0 + c1[xt]*p1   (pl_meta_model.py:125-137; host computes them in fp32)
   *   post[4] = 1 if target_t > 0 (Bernoulli draw) else 0 (clamp(min=0), :139-142)
   * gaussian : x_s = post[0]*(xt - post[1]*pred) + post[2]*pred + post[3]*z  (:161-172) */
  float post[8];
  int32_t rand_mode;      /* DIFUSCO_RAND_* */
  const float* rand;      /* injected uniforms / normals, caller order, or NULL */
  uint64_t seed, offset;  /* Philox key / per-call offset */

  /* outputs (device) */
  float* xt_out;          /* same shape/order as xt */
  float* pred_out;        /* optional: logits [rows,2] (categorical) or eps [rows] (gaussian) */
  float* prob_out;        /* optional, categorical: p(x_s=1) before clamping/sampling, [rows] */

  void* workspace;        /* device, >= difusco_workspace_bytes(...) */
  size_t workspace_bytes;
  void* stream;           /* hipStream_t */
  int32_t precision;      /* DIFUSCO_PREC_*: arithmetic of the E-row linears (node rows stay exact fp32) */
  int32_t no_fusion;      /* 0: use the fused edge-layer kernel when available (hidden == 256 and precision
                             BF16X3 / FP16X3); 1: always run the unfused kernel sequence (A/B and parity tests) */
  const int32_t* row;     /* [n_edges] centre node of each CSR slot (device); required for the fused kernel,
                             may be NULL otherwise */
  /* Head GroupNorm statistics across several calls (one per GPU of a sharded batch), SURVEY 8(e) option "global
   * statistics" = the reference's single call over all graphs.  gn_phase 0: the whole step in one call (statistics
   * of THIS call's rows).  1: run up to the statistics, write the 32 x (sum, sum of squares) of this call's rows
   * and the row count to gn_sums[0..63], gn_sums[64] (device doubles) and return; the caller adds gn_sums over the
   * shards (e.g. one all-reduce of 65 doubles).  2: finish the step of the preceding phase-1 call (same arguments,
   * same workspace) with the statistics in gn_sums.  Needs n_segments == 1. */
  int32_t gn_phase;
  int32_t flags;          /* DIFUSCO_FLAG_*: per-call A/B switches of the fused path (results equal to rounding) */
  double* gn_sums;
  /* Optional PREPARED STATE (ABI 9).  Both NULL = the stateless step: everything is computed from the arguments above.
   * A sampling loop calls the step 50 times with the same weights, graph and coordinates (pl_tsp_model.py:207-217); what
   * does not depend on x_t or t is then computed ONCE and handed back in:
   *   prepared  (TSP, fused path): the buffer difusco_prepare() filled for THIS (weights, precision, graph, points): the
   *             node embedding h0 = node_embed(pos_embed(points)) (gnn_encoder.py:394), layer 0's U|V|A|B rows of it and the
   *             two-row edge-input table of a categorical step - the step then skips those launches.  Ignored on the
   *             unfused path and for MIS (whose h0 is the embedding of x_t).
   *   tbias     [n_layers, hidden] time-bias rows of THIS t (difusco_time_bias_rows(); gnn_encoder.py:396,329-337,442) - the
   *             step skips the time MLP.
   * Same kernels, same operands: a step with prepared state is bit-identical to the stateless one (GPU test). */
  const void* prepared;
  const float* tbias;
  /* ABI 10.  DIFUSCO_AGG_*: h_i = U h_i + Aggr_j(gate_ij * V h_j) over the edges of centre node i (gnn_encoder.py:115,144-191).
   * SUM: torch_sparse.sum / torch.sum(dim=2).  MEAN: the sum divided by the number of edges of the row (torch_sparse.mean =
   * segment mean; the dense layer divides by sum(ones) = V).  MAX: the maximum over the row's edges (torch_sparse.max /
   * torch.max(dim=2)[0]); an empty row aggregates to 0 in all three.  MEAN runs the fused layers (the division happens where the
   * pieces of a row are added); MAX has fused instantiations of its own (the per-tile pieces of a row are maxima); only a MAX call
   * with n_nodes >= 2^20 takes the unfused kernel sequence. */
  int32_t aggregation;
  int32_t reserved0;      /* 0 */
  /* ABI 12.  Optional GENERATED-INPUT TABLE (TSP, fused path, hidden 256): the buffer difusco_gen_table_build() filled for THESE weights.
   * A step whose edge input is not a two-row lookup (Gaussian diffusion; a categorical x_t that is not exactly {0,1}) computes
   * e0 = edge_embed(ScalarEmbeddingSine(x_t)) (gnn_encoder.py:230-249,:304,:395): a function of ONE scalar per edge.  With the table,
   * every 32-edge tile whose x_t all lie in [-8, 8) is evaluated by degree-7 interpolation of eight sampled rows (interpolation error
   * 2e-9, far below the fp32 rounding of the contraction it replaces; csrc/edge_embed.hip) instead of 64 sincos + a K = 256 contraction
   * per edge; tiles with an edge outside the table or a non-finite x_t, and NULL, take the contraction.  Results differ from the
   * table-free step by fp32 rounding only. */
  const float* gen_table;
} difusco_step_args;

enum {
  DIFUSCO_FLAG_NO_L0_FOLD = 1,   /* first layer: write e0 to memory and run the general kernel (no 2-row table fold) */
  DIFUSCO_FLAG_NO_TAIL_FOLD = 2, /* last layer: general kernel + separate GroupNorm statistics pass over e */
  DIFUSCO_FLAG_CHECK_FINITE = 4  /* debugging aid: after the step, count the non-finite values of xt_out, of the head's GroupNorm
                                    statistics (nan as soon as the final state holds an inf / nan) and of pred_out / prob_out
                                    when given, synchronise the stream and return DIFUSCO_ENONFINITE if there are any.  The
                                    arithmetic has no operand-range restriction (any finite fp32 weights / inputs), so inf / nan
                                    can only come in with the inputs or from genuine fp32 overflow of the network itself. */
};

size_t difusco_workspace_bytes(int hidden, int n_layers, int n_nodes, int n_edges, int n_segments);

/* ---- prepared state (optional, see difusco_step_args.prepared / .tbias) -------------------------------------------
 * difusco_prepared_bytes(): size of the buffer for a call shape.  difusco_prepare(): fills it from args->{hidden, n_layers,
 * out_channels, task (TSP), weights, precision, n_nodes, n_edges, points, workspace, stream} - the graph arrays, x_t and
 * the outputs are not read.  `points` must be in the node numbering of the graph arrays the later steps pass.
 * difusco_time_bias_rows(): out[i] = the [n_layers, hidden] rows of t_host[i] (HOST array, n_t values), DEVICE
 * out [n_t, n_layers, hidden]; one launch per 64 values.  All asynchronous on the stream. */
size_t difusco_prepared_bytes(int hidden, int n_nodes);
int difusco_prepare(const difusco_step_args* args, void* prepared, size_t prepared_bytes);
int difusco_time_bias_rows(int hidden, int n_layers, int out_channels, const float* weights, const float* t_host, int n_t,
                           float* out, void* stream);

/* e0 = edge_embed(ScalarEmbeddingSine(x_t)) (difusco/models/gnn_encoder.py:230-249 the features, :304,:395 the linear) for a
 * GENERAL x_t (Gaussian diffusion; a categorical x_t that is not exactly {0,1}) - the kernel the fused step runs for such inputs,
 * exported for parity tests.  hidden = 256; precision = DIFUSCO_PREC_BF16X3 | _FP16X3.  xt [n_edges] in caller order, perm (or
 * NULL) maps CSR slot -> caller index; e_tiled: the TILED edge state (layout at difusco_edge_layer_fused below), padded to a multiple of 256 rows
 * (only the n_edges real rows are written); tile_max (or NULL): one float per 32-edge tile of the padded range = max |e0| of the
 * tile (0 for tiles past the end); gen_table (or NULL): the table of difusco_gen_table_build - tiles inside its range interpolate
 * (stand-alone entry: at most 581,632 edges per call with a table; the tile flags live in the table buffer's scratch). */
int difusco_edge_embed(int hidden, int n_layers, int out_channels, const float* weights, int precision, const float* xt,
                       const int32_t* perm, int64_t n_edges, float* e_tiled, float* tile_max, const float* gen_table, void* stream);

/* The generated-input table of difusco_step_args.gen_table (ABI 12): 71 rows of e0(x) at x = -8 + (r - 3) / 4, computed with the exact
 * fp32 kernels (precise sin / cos features, fp32-MFMA linear) from the blob's edge_embed weights; depends on the weights only - build once
 * per blob.  difusco_gen_table_bytes(): size of the buffer (rows + build scratch); hidden must be 256.  Asynchronous on the stream. */
size_t difusco_gen_table_bytes(int hidden);
int difusco_gen_table_build(int hidden, int n_layers, int out_channels, const float* weights, float* table, size_t table_bytes,
                            void* stream);

/* One reverse-diffusion step: GNN denoiser forward + posterior (+ sample).  Asynchronous on
 * args->stream.  Replaces {categorical,gaussian}_denoise_step of pl_tsp_model.py / pl_mis_model.py. */
int difusco_denoise_step(const difusco_step_args* args);

/* ---- single kernels, exported for parity tests and profiling ------------------------------------ */
/* Y[m, ldy] (cols [0,n_out)) = X[m,k] * W[n_out,k]^T + bias (+ residual, same layout as Y).
 * k in {32,64,128,256}; n_out multiple of 32.  fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int difusco_linear_rows(const float* x, const float* w, const float* bias, const float* residual,
                        float* y, int64_t m, int k, int n_out, int64_t ldy, void* stream);
/* Same contract on the split-precision path: `planes` = the five 16-bit planes of W[n_out,k] in the
 * *_PLANES layout above (planes + inverse scales); precision = DIFUSCO_PREC_BF16X3 | _BF16X6 | _FP16X3.
 * k == n_out in {64,128,256}, or k = 256 with n_out a multiple of 256 (the node-row shape, n_out = 1024: a row-major y with
 * ldy == n_out and no residual then takes the register-resident kernel of node_linear.hip).  row_scale_scratch (device, m floats, or NULL): with FP16X3 the rows of x are scaled
 * individually by a power of two computed in an extra pass (any finite |x|); NULL skips that pass and then needs
 * 2^-3 <= |x| < 65504 for the full 22 bits. */
int difusco_linear_rows_split(const float* x, const void* planes, int precision, const float* bias,
                              const float* residual, float* y, int64_t m, int k, int n_out, int64_t ldy,
                              float* row_scale_scratch, void* stream);

/* One gated-GCN message-passing pass (gnn_encoder.py:110-135 + :445-448 + per_layer_out LN/SiLU):
 *   e' = Ah[j]+Bh[i]+Ce ; h[i] += ReLU(LN_h(Uh[i] + sum_j sigmoid(e')*Vh[j])) (+tbias, MIS)
 *   ce_act[s] <- SiLU(LN_o(ReLU(LN_e(e')) (+tbias, TSP)))      (in place over Ce)
 * node4 = [n_nodes,4H] rows U|V|A|B.  ln = {norm_h w,b, norm_e w,b, out_ln w,b} device pointers. */
int difusco_edge_gate_aggregate(int hidden, int n_nodes, const int32_t* rowptr, const int32_t* col,
                                const float* node4, float* ce_act, float* h,
                                const float* norm_h_w, const float* norm_h_b,
                                const float* norm_e_w, const float* norm_e_b,
                                const float* out_ln_w, const float* out_ln_b,
                                const float* tbias, int time_on_edge, void* stream);

/* The fused edge pass of one layer + node update (edge_layer.hip), H = 256 only:
 *   e <- e + W_o SiLU(LN_o(ReLU(LN_e(Ah[j]+Bh[i]+C e)) (+t))) + b_o ;  h_i += ReLU(LN_h(Uh_i + sum gate*Vh_j)) (+t)
 * e is in the TILED layout of the fused path: rows padded to a multiple of 256 edges (pad = 0), 32-edge
 * tiles of 8192 floats ordered [f/16][(f/8)%2][((f/4)%2)*32 + s%32][f%4] (csrc/kernels.h edge_tiled_offset,
 * difusco_amd.graph.to_tiled) - every wavefront access is then one contiguous KiB.
 * planes_c / planes_o: the five 16-bit planes of C / per_layer_out[l][2] (see *_PLANES above);
 * precision = DIFUSCO_PREC_BF16X3 | DIFUSCO_PREC_FP16X3.  scales: the layer's DIFUSCO_WL_FUSED_SCALES record (device,
 * 8 floats; required for FP16X3, ignored for BF16X3).  scratch: >= difusco_fused_scratch_bytes().
 * node4 is the REFERENCE's U h | V h | A h | B h (gnn_encoder.py:94-103); this entry converts the A | B rows into the kernel's
 * log2(e) domain itself (a copy in the scratch, ABI 11) - the step driver has no such pass, its node linear writes them so. */
size_t difusco_fused_scratch_bytes(int n_nodes, int n_edges);
int difusco_edge_layer_fused(int precision, int n_nodes, int n_edges, const int32_t* rowptr, const int32_t* row,
                             const int32_t* col, const float* node4, float* e, float* h, const void* planes_c,
                             const void* planes_o, const float* b_c, const float* norm_h_w, const float* norm_h_b,
                             const float* norm_e_w, const float* norm_e_b, const float* out_ln_w,
                             const float* out_ln_b, const float* b_out, const float* tbias, int time_on_edge,
                             const float* scales, void* scratch, void* stream);

/* Elementwise posteriors on already computed predictions (pl_meta_model.py:102-175). */
int difusco_categorical_posterior(const float* logits, const float* xt, const float* post,
                                  int rand_mode, const float* rand, uint64_t seed, uint64_t offset,
                                  float* xt_out, float* prob_out, int64_t n, void* stream);
int difusco_gaussian_posterior(const float* pred, const float* xt, const float* post,
                               int rand_mode, const float* rand, uint64_t seed, uint64_t offset,
                               float* xt_out, int64_t n, void* stream);

/* ---- k-NN graph in the reference's layout: co_datasets/tsp_graph_dataset.py:53-62
 * (sklearn KDTree(points).query(points, k) on float64 coordinates).  points: DEVICE float64 [n_nodes,2]; writes
 * edge_row0[i*k + r] = node_offset + i and edge_row1[i*k + r] = node_offset + (r-th nearest neighbour of i, self
 * first, ties by lower index) - DEVICE int64, i.e. the graph's slice of a [2, E_tot] edge_index of a disjoint-union
 * batch (pl_meta_model.py:177-184).  1 <= k <= min(n_nodes, 1024).  Asynchronous on `stream`. */
int difusco_knn_graph_workspace_bytes(int n_nodes, int k, size_t* bytes);
int difusco_knn_graph(int n_nodes, int k, const double* points, int64_t node_offset, int64_t* edge_row0,
                      int64_t* edge_row1, void* workspace, size_t workspace_bytes, void* stream);

/* ---- greedy MIS decode: difusco/utils/mis_utils.py:3-18 (mis_decode_np).  rowptr/col: DEVICE CSR of
 * the symmetric adjacency of the whole call (self loops allowed, as in co_datasets/mis_dataset.py:43-48; the
 * graphs of a batch are independent components); scores: DEVICE float32 [n_nodes] (the final x_t, + 1e-6 or
 * * 0.5 + 0.5 applied); solution: DEVICE int32 [n_nodes], 1 = in the independent set.  Visiting order = decreasing
 * score, equal scores by increasing node id.  *rounds_out (HOST, optional) = parallel rounds used.  Blocks. */
int difusco_mis_decode_workspace_bytes(int n_nodes, size_t* bytes);
int difusco_mis_decode(int n_nodes, const int32_t* rowptr, const int32_t* col, const float* scores, int32_t* solution,
                       void* workspace, size_t workspace_bytes, int32_t* rounds_out, void* stream);

/* ---- heatmap -> tour: the greedy edge insertion the reference runs on the host right after the
 * sampling loop, difusco/utils/tsp_utils.py:89-145 (merge_tours) + utils/cython_merge/cython_merge.pyx:19-104
 * (merge_cython), restricted to the E entries of the sparse heatmap instead of the N x N densification.
 * ONE graph per call, node ids 0..n_nodes-1 (the caller slices a batch).  row/col/heat/points/workspace are DEVICE
 * pointers: row[e] -> col[e] are the directed edges of the sparse graph in any order, heat[e] the final x_t
 * (+1e-6 / *0.5+0.5 already applied, pl_tsp_model.py:219-222), points [n_nodes,2] float32.  tour_out (HOST,
 * n_nodes + 1 ints) receives the closed tour starting and ending at node 0 (tsp_utils.py:134-141);
 * *merge_iterations the reference's counter over its dense sorted list; *completed = 1 when the N-1 insertions
 * were found among candidate pairs with a positive score (the regime in which the result is pinned to the
 * reference), 0 when the fallback had to finish the tour.  Blocks until done (synchronises `stream`). */
int difusco_tsp_merge_workspace_bytes(int64_t n_edges, size_t* bytes);
int difusco_tsp_merge_tour(int n_nodes, int64_t n_edges, const int32_t* row, const int32_t* col, const float* heat,
                           const float* points, void* workspace, size_t workspace_bytes, int32_t* tour_out,
                           int64_t* merge_iterations, int32_t* completed, void* stream);
/* The `parallel_sampling` samples of ONE graph in one call (the loop of tsp_utils.py:100-145 over adj_mat[i]):
 * heat [n_samples][n_edges] DEVICE, tours_out [n_samples][n_nodes + 1] HOST, merge_iterations / completed [n_samples]
 * HOST (optional).  The pair keys depend on the graph only, so their sort runs once per call; the workspace is the
 * one of difusco_tsp_merge_workspace_bytes.  Results per sample equal difusco_tsp_merge_tour's. */
int difusco_tsp_merge_tours(int n_nodes, int64_t n_edges, const int32_t* row, const int32_t* col, const float* heat,
                            const float* points, int n_samples, void* workspace, size_t workspace_bytes,
                            int32_t* tours_out, int64_t* merge_iterations, int32_t* completed, void* stream);

/* ---- batched 2-opt: difusco/utils/tsp_utils.py:12-49 (batched_two_opt_torch).  points: DEVICE
 * float64 [n_nodes,2]; tours: DEVICE int32 [batch, n_nodes+1] closed tours, refined in place.  Every iteration
 * finds, per tour, the move (i,j), j >= i+2, that minimises d(t_i,t_j) + d(t_i+1,t_j+1) - d(t_i,t_i+1) - d(t_j,t_j+1)
 * (float64, the reference's operation order, first flat index on ties) and reverses tour[i+1..j]; it stops when the
 * best change over the whole batch is >= -1e-6 or after max_iterations applied moves.  *iterations_out (HOST) =
 * the reference's `iterator`.  Blocks until done. */
int difusco_tsp_two_opt_workspace_bytes(int n_nodes, int batch, size_t* bytes);
int difusco_tsp_two_opt(int n_nodes, int batch, const double* points, int32_t* tours, int64_t max_iterations,
                        void* workspace, size_t workspace_bytes, int64_t* iterations_out, void* stream);

/* ---- MCTS heatmap rows: the numeric part of tsp_mcts/convert_numpy_to_txt.py:18-47, whose text output
 * (first line N, then N rows of N "%.6f" numbers) tsp_mcts/code/include/TSP_IO.h:461-492 reads.  From the SPARSE heatmap:
 * row/col/heat [n_edges] DEVICE, any order, no duplicate (row, col); points DEVICE float32 [n_nodes,2]; float32 arithmetic
 * in numpy's operation order (no fused multiply-add, numpy's chunked pairwise row sums), so that the rows equal the
 * reference program's bit for bit.  prepare(): CSR of heat and its transpose, the threshold (k-th largest positive value,
 * k = int(N*N*prob), exact radix select; k = 0 selects the smallest positive value like the reference's
 * valid_values[-0]) and the row top-3 - kept in `workspace`; *threshold_out (HOST, optional).  rows(): the normalised
 * rows [row_begin, row_begin + row_count) into out_rows (DEVICE, [row_count, n_nodes]).  n_nodes <= 38000 (a row lives
 * in LDS).  Both block until done. */
/* HOST helper (no GPU): the float32 sum of a[0..n) in the order of the row-sum program the kernels execute (numpy's
 * add.reduce over a contiguous float32 row: 8192-element chunks, pairwise summation inside) - exported so that the CPU
 * tests can hold that program against numpy itself. */
int difusco_host_rowsum_f32(const float* a, int n, float* out);
int difusco_mcts_heatmap_workspace_bytes(int n_nodes, int64_t n_edges, size_t* bytes);
int difusco_mcts_heatmap_prepare(int n_nodes, int64_t n_edges, const int32_t* row, const int32_t* col, const float* heat,
                                 const float* points, double expected_valid_prob, void* workspace, size_t workspace_bytes,
                                 float* threshold_out, void* stream);
int difusco_mcts_heatmap_rows(int n_nodes, int64_t n_edges, const float* points, const void* workspace,
                              size_t workspace_bytes, int row_begin, int row_count, float* out_rows, void* stream);

/* ---- in-library profiler (bench.py): HIP events on the launch stream around every kernel launch of
 * difusco_denoise_step, summed per category.  Categories: 0 edge-row linear (rows = n_edges),
 * 1 node-row linear, 2 edge gate/aggregate, 3 head (GroupNorm+conv+posterior, 3 launches),
 * 4 embeddings / time features / memset.  enable(1, max) (re)arms and pre-creates the events; enable(2, max)
 * brackets category 0 only (every bracket costs a few microseconds of dispatch gap: ~2.7 % of a step for all five);
 * collect() synchronises, fills ms[c] / launches[c] for c < 5, re-arms, returns brackets read. */
#define DIFUSCO_PROFILE_CATEGORIES 5
int difusco_profile_enable(int on, int max_launches);
int difusco_profile_collect(double* ms, int64_t* launches, int n_categories);

#ifdef DIFUSCO_PROFILING
/* Profiling library only (libdifusco_hip_prof.so, built with -DDIFUSCO_PROFILING; `python -m difusco_amd.build --prof`):
 * process-wide knobs that select timing-only variants of the fused edge-layer kernel.  The production library does
 * not export these symbols and holds no such state.
 * key 0: compile-time ablation mask (bit0 skip neighbour-table gathers, bit1 skip the neighbour sum, bit2 skip
 *        LN/activation math, bit3 skip GEMM 2, 16 = production code + phase stamps ...): results are WRONG by design;
 * key 6: extra dynamic LDS bytes (occupancy probe);  key 7: A/B variant of the scheduling options (OPT bits);
 * key 8: k steps of load lookahead in the node-row linear (0, 1 or 4);  key 9: start delay (cycles) of the second-slot workgroups of
 *        the fused kernel's first dispatch generation (experiment of round 3);  key 10: timing-only ablation mask of the node-row
 *        linear (node_linear.hip, 0..31). */
int difusco_debug_set(int key, int value);
/* key 1: device buffer [n_tiles][16] of uint64 receiving s_memtime stamps of the fused kernel's phases (NULL disables) */
int difusco_debug_set_ptr(int key, void* p);
/* Stage-loop laboratory (csrc/stage_lab.hip): GEMM 1 of the fused edge layer alone - out = C e on the tiled e stream
 * (fp16 hi | lo planes of C, 3 MFMA products, weight stages through LDS by LDS-DMA) - in the workgroup geometry /
 * synchronisation scheme `variant` names (EPW/32 * 100000 + WAVES * 10000 + NBUF * 1000 + SYNC * 100 + RING * 10 + PRIO).
 * e / out: DEVICE, tiled [n_edges, 256] (n_edges a multiple of the variant's edges per workgroup); inv_c = 2^-kc of the
 * planes; do_store 0 = timing only; lds_pad = extra dynamic LDS bytes.  scripts/bench_stage_lab.py drives it. */
int difusco_lab_gemm1(int variant, const float* e, const void* planes, float* out, int n_edges, float inv_c, int do_store,
                      int lds_pad, void* stream);
/* Re-read probe: one streaming pass (float4 loads, non-temporal or default policy) over buf[0 .. n_floats); scripts/
 * bench_reread_probe.py times a second pass over the same buffer against the first for several sizes. */
int difusco_lab_reread_pass(const float* buf, long long n_floats, int nontemporal, float* sink, void* stream);
/* the same translation unit compiled without packed fp32 arithmetic (target feature -packed-fp32-ops) */
int difusco_lab_gemm1_nopk(int variant, const float* e, const void* planes, float* out, int n_edges, float inv_c, int do_store,
                           int lds_pad, void* stream);
#endif

#ifdef __cplusplus
}
#endif
#endif /* DIFUSCO_HIP_H */
