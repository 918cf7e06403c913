// HBM-bound kernels of the DIFUSCO denoise step for gfx950: embeddings, the CSR edge-gate /
// neighbour-aggregation pass, the GroupNorm head and the reverse-diffusion posteriors.
//
// Feature layout everywhere: a [rows, H] fp32 row is spread over the 64 lanes of ONE wavefront,
// lane l owning the VEC = H/64 consecutive channels [l*VEC, (l+1)*VEC) (H=256 -> one float4 per lane,
// 1 KiB per row per wave instruction, fully coalesced).  Reductions over H (LayerNorm, the 1x1 conv)
// are wavefront butterflies; the neighbour sum over a node's edges is a per-lane register
// accumulation by the wave that owns the node (deterministic order, no atomics).
#include "common.h"
#include "kernels.h"

namespace difusco {

template <int VEC>
struct Vec {
  float v[VEC];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> ldv(const float* p) {
  Vec<VEC> r;
  if constexpr (VEC == 4) {
    const v4f t = *reinterpret_cast<const v4f*>(p);
    r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
  } else if constexpr (VEC == 2) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f t = *reinterpret_cast<const v2f*>(p);
    r.v[0] = t[0]; r.v[1] = t[1];
  } else {
    r.v[0] = *p;
  }
  return r;
}

template <int VEC>
__device__ __forceinline__ void stv(float* p, const Vec<VEC>& r) {
  if constexpr (VEC == 4) {
    v4f t;
    t[0] = r.v[0]; t[1] = r.v[1]; t[2] = r.v[2]; t[3] = r.v[3];
    *reinterpret_cast<v4f*>(p) = t;
  } else if constexpr (VEC == 2) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f t;
    t[0] = r.v[0]; t[1] = r.v[1];
    *reinterpret_cast<v2f*>(p) = t;
  } else {
    *p = r.v[0];
  }
}

// LayerNorm over the H = 64*VEC channels held by one wavefront (eps 1e-5, biased variance,
// torch.nn.LayerNorm semantics; gnn_encoder.py:58-65).  Two-pass on registers.
template <int VEC>
__device__ __forceinline__ Vec<VEC> wave_layer_norm(const Vec<VEC>& x, const Vec<VEC>& gamma, const Vec<VEC>& beta) {
  constexpr float inv_h = 1.0f / (64 * VEC);
  float s = 0.0f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) s += x.v[v];
  const float mean = wave_sum(s) * inv_h;
  float q = 0.0f;
  Vec<VEC> d;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    d.v[v] = x.v[v] - mean;
    q += d.v[v] * d.v[v];
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_h + 1e-5f);
  Vec<VEC> y;
#pragma unroll
  for (int v = 0; v < VEC; ++v) y.v[v] = d.v[v] * rstd * gamma.v[v] + beta.v[v];
  return y;
}

// ------------------------------------------------------------------------------------------------
// time features: timestep_embedding -> time_embed MLP -> per-layer ReLU+Linear
// (models/nn.py:103-121, gnn_encoder.py:311-315, :329-337).  One workgroup per layer; one [1,H]
// row per layer per step, so this is latency-, not bandwidth-relevant.
// ------------------------------------------------------------------------------------------------
struct TimeBatch {      // up to 64 diffusion times per launch (blockIdx.y selects one)
  float t[64];
};
__global__ __launch_bounds__(256) void time_bias_kernel(TimeBatch tb, int H, const float* __restrict__ freqs,
                                                        const float* __restrict__ w0, const float* __restrict__ b0,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        const float* __restrict__ wl_base, long long layer_stride,
                                                        long long wl_w_off, long long wl_b_off, float* __restrict__ tbias) {
  __shared__ float emb[256];
  __shared__ float h1[128];
  __shared__ float te[128];
  const int half = H / 2;
  const int tid = threadIdx.x;
  const float t = tb.t[blockIdx.y];
  tbias += (long long)blockIdx.y * gridDim.x * H;      // rows of time blockIdx.y: [n_layers, H]
  for (int k = tid; k < half; k += 256) {
    const float a = t * freqs[k];
    emb[k] = cosf(a);
    emb[half + k] = sinf(a);
  }
  __syncthreads();
  for (int o = tid; o < half; o += 256) {
    float s = 0.0f;
    for (int k = 0; k < H; ++k) s += w0[o * H + k] * emb[k];
    s += b0[o];
    h1[o] = s > 0.0f ? s : 0.0f;
  }
  __syncthreads();
  for (int o = tid; o < half; o += 256) {
    float s = 0.0f;
    for (int k = 0; k < half; ++k) s += w2[o * half + k] * h1[k];
    s += b2[o];
    te[o] = s > 0.0f ? s : 0.0f;  // the ReLU that opens every time_embed_layers[l]
  }
  __syncthreads();
  const int l = blockIdx.x;
  const float* wl = wl_base + l * layer_stride + wl_w_off;
  const float* bl = wl_base + l * layer_stride + wl_b_off;
  for (int o = tid; o < H; o += 256) {
    float s = 0.0f;
    for (int k = 0; k < half; ++k) s += wl[o * half + k] * te[k];
    tbias[l * H + o] = s + bl[o];
  }
}

// ------------------------------------------------------------------------------------------------
// sinusoidal input embeddings (inputs of node_embed / edge_embed)
// ------------------------------------------------------------------------------------------------
// PositionEmbeddingSine(H/2, normalize=True): gnn_encoder.py:194-227.  out[n, 0:H/2] from coord 0,
// out[n, H/2:H] from coord 1; even channel sin, odd channel cos.
__global__ void pos_embed_kernel(const float* __restrict__ points, const float* __restrict__ dimt, int n_nodes, int H,
                                 float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n_nodes * H) return;
  const int c = (int)(idx % H);
  const long long n = idx / H;
  const int half = H / 2;
  const int coord = c < half ? 0 : 1;
  const int k = c < half ? c : c - half;
  const float v = (points[n * 2 + coord] * 6.283185307179586f) / dimt[k];
  out[idx] = (k & 1) ? cosf(v) : sinf(v);
}

// ScalarEmbeddingSine / ScalarEmbeddingSine1D: gnn_encoder.py:230-271.  x index through `perm`
// (CSR slot -> caller order) when given.
__global__ void scalar_embed_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                    const float* __restrict__ dimt, long long rows, int H, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * H) return;
  const int c = (int)(idx % H);
  const long long r = idx / H;
  const float xv = x ? x[perm ? perm[r] : r] : (float)r;  // x == nullptr: rows are the constants 0,1,...
  const float v = xv / dimt[c];
  out[idx] = (c & 1) ? cosf(v) : sinf(v);
}

// categorical inference: xt is exactly 0/1, so edge_embed(ScalarEmbeddingSine(xt)) has two distinct
// rows; table = those two rows (computed by the same embed+linear kernels on x = {0,1}).
template <int VEC>
__global__ __launch_bounds__(256) void table_rows_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                         const float* __restrict__ table, long long rows,
                                                         float* __restrict__ out) {
  constexpr int H = 64 * VEC;
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const Vec<VEC> r0 = ldv<VEC>(table + lane * VEC);
  const Vec<VEC> r1 = ldv<VEC>(table + H + lane * VEC);
  for (long long r = wave0; r < rows; r += nwaves) {
    const float xv = x[perm ? perm[r] : r];
    stv<VEC>(out + r * H + lane * VEC, xv > 0.5f ? r1 : r0);
  }
}

// ------------------------------------------------------------------------------------------------
// CSR edge gate + neighbour aggregation + the two LayerNorms on the edge row
//   e'    = Ah[j] + Bh[i] + Ce                         gnn_encoder.py:110
//   agg_i = sum_j sigmoid(e') * Vh[j]                  :112,:115,:163,:177-191 (aggregation='sum'; agg_mode 1 'mean': the sum
//           divided by the row length, 2 'max': the maximum over the row's edges, 0 for an empty row - :170-173,184-188,
//           torch_sparse.mean / max = torch_scatter segment_csr semantics)
//   h_i  += ReLU(LN_h(Uh[i] + agg_i)) (+ tbias, MIS)   :123,:134,:447-448
//   act   = SiLU(LN_o(ReLU(LN_e(e')) (+ tbias, TSP)))  :131,:135,:445, per_layer_out[l][0:2] :339-342
// One wavefront owns one centre node i and walks its CSR row (latency of the four dependent wave
// reductions per edge is hidden by the other resident waves: the kernel needs few registers).
// node4 = [n_nodes, 4H] with rows U|V|A|B.  ce_act: in = C e + bias, out = act (same slot, same lane).
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void edge_gate_aggregate_kernel(
    int n_nodes, const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ node4,
    float* ce_act, float* h, const float* __restrict__ nh_w, const float* __restrict__ nh_b,
    const float* __restrict__ ne_w, const float* __restrict__ ne_b, const float* __restrict__ ol_w,
    const float* __restrict__ ol_b, const float* __restrict__ tbias, int time_on_edge, int agg_mode) {
  constexpr int H = 64 * VEC;
  const int lane = threadIdx.x & 63;
  const int f = lane * VEC;
  // wave-uniform node id, made provably uniform so row pointers / neighbour ids are scalar loads
  const int i = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (i >= n_nodes) return;

  const float* ni = node4 + (long long)i * 4 * H;
  const Vec<VEC> uh = ldv<VEC>(ni + f);
  const Vec<VEC> bh = ldv<VEC>(ni + 3 * H + f);
  const Vec<VEC> g_e = ldv<VEC>(ne_w + f), b_e = ldv<VEC>(ne_b + f);
  const Vec<VEC> g_o = ldv<VEC>(ol_w + f), b_o = ldv<VEC>(ol_b + f);
  const Vec<VEC> tb = ldv<VEC>(tbias + f);

  const int s_begin = rowptr[i], s_end = rowptr[i + 1];
  const bool agg_max = agg_mode == 2 && s_end > s_begin;      // (wave uniform)
  Vec<VEC> agg;
#pragma unroll
  for (int v = 0; v < VEC; ++v) agg.v[v] = agg_max ? -__builtin_inff() : 0.0f;

  for (int s = s_begin; s < s_end; ++s) {
    const int j = col[s];
    const float* nj = node4 + (long long)j * 4 * H;
    const Vec<VEC> vh = ldv<VEC>(nj + H + f);
    const Vec<VEC> ah = ldv<VEC>(nj + 2 * H + f);
    float* cp = ce_act + (long long)s * H + f;
    const Vec<VEC> ce = ldv<VEC>(cp);
    Vec<VEC> e;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      e.v[v] = (ah.v[v] + bh.v[v]) + ce.v[v];
      if (agg_max) agg.v[v] = __builtin_fmaxf(agg.v[v], sigmoidf_(e.v[v]) * vh.v[v]);      // (wave uniform)
      else agg.v[v] += sigmoidf_(e.v[v]) * vh.v[v];
    }
    Vec<VEC> y = wave_layer_norm<VEC>(e, g_e, b_e);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      y.v[v] = y.v[v] > 0.0f ? y.v[v] : 0.0f;
      if (time_on_edge) y.v[v] += tb.v[v];
    }
    Vec<VEC> z = wave_layer_norm<VEC>(y, g_o, b_o);
#pragma unroll
    for (int v = 0; v < VEC; ++v) z.v[v] = z.v[v] * sigmoidf_(z.v[v]);
    stv<VEC>(cp, z);
  }

  if (agg_mode == 1 && s_end > s_begin) {
    const float cnt = (float)(s_end - s_begin);
#pragma unroll
    for (int v = 0; v < VEC; ++v) agg.v[v] = agg.v[v] / cnt;
  }
  Vec<VEC> hn;
#pragma unroll
  for (int v = 0; v < VEC; ++v) hn.v[v] = uh.v[v] + agg.v[v];
  const Vec<VEC> g_h = ldv<VEC>(nh_w + f), b_h = ldv<VEC>(nh_b + f);
  hn = wave_layer_norm<VEC>(hn, g_h, b_h);
  float* hp = h + (long long)i * H + f;
  Vec<VEC> hi = ldv<VEC>(hp);
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float r = hn.v[v] > 0.0f ? hn.v[v] : 0.0f;
    if (!time_on_edge) r += tb.v[v];
    hi.v[v] += r;
  }
  stv<VEC>(hp, hi);
}

// ------------------------------------------------------------------------------------------------
// head: GroupNorm32(32,H) statistics over (H/32 channels x all rows of a segment)
// (gnn_encoder.py:316-322,400-401,412-413; nn.py:17-19,93-100).  A group = 2 adjacent lanes.
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ feat, const int* __restrict__ seg_ptr,
                                                         long long total_rows, double* __restrict__ partial) {
  constexpr int H = 64 * VEC;
  __shared__ double red[4][32][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = blockIdx.y;
  const long long r_begin = seg_ptr ? seg_ptr[seg] : 0;
  const long long r_end = seg_ptr ? seg_ptr[seg + 1] : total_rows;
  double s = 0.0, q = 0.0;
  for (long long r = r_begin + blockIdx.x * 4 + wave; r < r_end; r += (long long)gridDim.x * 4) {
    const Vec<VEC> x = ldv<VEC>(feat + r * H + lane * VEC);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s += (double)x.v[v];
      q += (double)x.v[v] * (double)x.v[v];
    }
  }
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  if ((lane & 1) == 0) {
    red[wave][lane >> 1][0] = s;
    red[wave][lane >> 1][1] = q;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 1, w = threadIdx.x & 1;
    const double t = red[0][g][w] + red[1][g][w] + red[2][g][w] + red[3][g][w];
    partial[((long long)seg * gridDim.x + blockIdx.x) * 64 + g * 2 + w] = t;
  }
}

// stats[seg][g] = {mean, rstd}; one 256-thread block per segment: thread (c, g, w) adds every 4th block
// partial of moment w of group g, the 4 chains are combined in a fixed order (deterministic).
// group_stride_blocks: 1 when every block partial holds all 32 groups (row-major kernels); 8 when block b only
// holds groups 4 (b % 8) .. 4 (b % 8) + 3 (tiled kernel).
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const double* __restrict__ partial, const int* __restrict__ seg_ptr,
                                                          long long total_rows, int nblk, int ch_per_group,
                                                          int group_stride_blocks, float* __restrict__ stats,
                                                          double* __restrict__ sums_out) {
  // sums_out (one segment only): instead of the statistics, write the 32 x (sum, sum of squares) and the row count
  // [64] - the quantities a caller adds over the shards of a batch (difusco_step_args.gn_phase 1)
  // 16 chains (wavefronts) per moment: each adds every 16th block partial, the chains are combined in a fixed order
  // (deterministic; 4 chains of 64 dependent loads made this kernel 19 us of pure latency)
  __shared__ double red[16][64];
  const int seg = blockIdx.x;
  const int c = threadIdx.x >> 6, gw = threadIdx.x & 63, g = gw >> 1, nch = blockDim.x >> 6;
  double acc = 0.0;
  if (group_stride_blocks == 1) {
    for (int b = c; b < nblk; b += nch) acc += partial[((long long)seg * nblk + b) * 64 + gw];
  } else {
    for (int b = (g >> 2) + 8 * c; b < nblk; b += 8 * nch) acc += partial[((long long)seg * nblk + b) * 64 + gw];
  }
  red[c][gw] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int gg = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nch; ++k) {
      s += red[k][2 * gg];
      q += red[k][2 * gg + 1];
    }
    const long long rows = seg_ptr ? (long long)(seg_ptr[seg + 1] - seg_ptr[seg]) : total_rows;
    if (sums_out) {
      sums_out[2 * gg] = s;
      sums_out[2 * gg + 1] = q;
      if (gg == 0) sums_out[64] = (double)rows;
      return;
    }
    const double cnt = (double)rows * ch_per_group;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[(seg * 32 + gg) * 2 + 0] = (float)mean;
    stats[(seg * 32 + gg) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
  }
}

// statistics from (possibly shard-summed) group sums: sums[2g] = sum, sums[2g+1] = sum of squares, sums[64] = rows
__global__ void gn_stats_from_sums_kernel(const double* __restrict__ sums, int ch_per_group, float* __restrict__ stats) {
  const int gg = threadIdx.x;
  if (gg >= 32) return;
  const double cnt = sums[64] * ch_per_group;
  const double mean = sums[2 * gg] / cnt;
  double var = sums[2 * gg + 1] / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[gg * 2 + 0] = (float)mean;
  stats[gg * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// ---- posterior arithmetic (shared by the fused head and the stand-alone kernels) -------------------
// categorical: pl_tsp_model.py:133-135 (softmax over the 2 classes) + pl_meta_model.py:125-142.
// post = {c0[xt=0], c0[xt=1], c1[xt=0], c1[xt=1], draw}.
struct PostParams {
  float p[8];
  int rand_mode;
  const float* rand;
  unsigned long long seed, offset;
};

// No multiply-add contraction in the two posteriors: the reference evaluates them as separate fp32 torch operations.  HIP's
// __fmul_rn / __fadd_rn are plain operators, hipcc's default -ffp-contract=fast fuses them in the backend whatever the source
// says (a `#pragma clang fp contract(off)` is not honoured in that mode: checked in the ISA) - round 4 found the Gaussian update
// off by one ulp of x_t on 0.6 % of the elements for that reason.  An empty asm that "modifies" the intermediate pins it.
__device__ __forceinline__ float rounded(float x) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the "v" constraint does not exist on the host pass)
  asm volatile("" : "+v"(x));
#endif
  return x;
}

__device__ __forceinline__ float categorical_step(float l0, float l1, float xt, const PostParams& pp, long long idx,
                                                  float* prob_out) {
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float den = e0 + e1;
  const float p0 = e0 / den, p1 = e1 / den;
  const int b = (int)xt >= 1 ? 1 : 0;      // x_t.long() (truncation, pl_meta_model.py:122); 0/1 inputs: the bit itself
  const float prob = rounded(pp.p[b] * p0) + rounded(pp.p[2 + b] * p1);
  if (prob_out) *prob_out = prob;
  if (pp.p[4] != 0.0f) {
    const float pc = fminf(fmaxf(prob, 0.0f), 1.0f);
    float u;
    if (pp.rand_mode == 1) u = pp.rand[idx];
    else u = philox_uniform(pp.seed, pp.offset, (unsigned long long)idx);
    return u < pc ? 1.0f : 0.0f;
  }
  return fmaxf(prob, 0.0f);
}

// gaussian: pl_meta_model.py:161-172.  post = {a, b, c, d, branch}: branch 0 (DDIM)
//   x = a*(xt - b*pred) + c*pred ; branch 1 (DDPM) x = a*(xt - b*pred) + d*z.
__device__ __forceinline__ float gaussian_step(float pred, float xt, const PostParams& pp, long long idx) {
  const float base = rounded(pp.p[0] * rounded(xt - rounded(pp.p[1] * pred)));
  if (pp.p[4] == 0.0f) return base + rounded(pp.p[2] * pred);
  float z;
  if (pp.rand_mode == 1) z = pp.rand[idx];
  else z = philox_normal(pp.seed, pp.offset, (unsigned long long)idx);
  return base + rounded(pp.p[3] * z);
}

// head apply: GroupNorm affine -> ReLU -> 1x1 conv (C = 1 or 2) -> softmax/posterior/sample.
// One wavefront per output row.  Outputs are written in CALLER order (through perm).
template <int VEC, int C>
__global__ __launch_bounds__(256) void head_apply_kernel(const float* __restrict__ feat, const int* __restrict__ seg_ptr,
                                                         long long total_rows, const float* __restrict__ stats,
                                                         const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                         const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                                         const int* __restrict__ perm, const float* __restrict__ xt,
                                                         PostParams pp, float* __restrict__ xt_out,
                                                         float* __restrict__ pred_out, float* __restrict__ prob_out) {
  constexpr int H = 64 * VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = blockIdx.y;
  const long long r_begin = seg_ptr ? seg_ptr[seg] : 0;
  const long long r_end = seg_ptr ? seg_ptr[seg + 1] : total_rows;
  const float mean = stats[(seg * 32 + (lane >> 1)) * 2 + 0];
  const float rstd = stats[(seg * 32 + (lane >> 1)) * 2 + 1];
  const Vec<VEC> gw = ldv<VEC>(gn_w + lane * VEC), gb = ldv<VEC>(gn_b + lane * VEC);
  Vec<VEC> cw[C];
#pragma unroll
  for (int c = 0; c < C; ++c) cw[c] = ldv<VEC>(conv_w + c * H + lane * VEC);
  for (long long r = r_begin + blockIdx.x * 4 + wave; r < r_end; r += (long long)gridDim.x * 4) {
    const Vec<VEC> x = ldv<VEC>(feat + r * H + lane * VEC);
    float dot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dot[c] = 0.0f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float y = (x.v[v] - mean) * rstd * gw.v[v] + gb.v[v];
      y = y > 0.0f ? y : 0.0f;
#pragma unroll
      for (int c = 0; c < C; ++c) dot[c] += y * cw[c].v[v];
    }
    if constexpr (C == 2) {
      wave_sum2(dot[0], dot[1]);
    } else {
      dot[0] = wave_sum(dot[0]);
    }
    if (lane == 0) {
      const long long idx = perm ? perm[r] : r;
      if constexpr (C == 2) {
        const float l0 = dot[0] + conv_b[0], l1 = dot[1] + conv_b[1];
        if (pred_out) {
          pred_out[idx * 2 + 0] = l0;
          pred_out[idx * 2 + 1] = l1;
        }
        xt_out[idx] = categorical_step(l0, l1, xt[idx], pp, idx, prob_out ? prob_out + idx : nullptr);
      } else {
        const float pred = dot[0] + conv_b[0];
        if (pred_out) pred_out[idx] = pred;
        xt_out[idx] = gaussian_step(pred, xt[idx], pp, idx);
      }
    }
  }
}

__global__ void categorical_posterior_kernel(const float* __restrict__ logits, const float* __restrict__ xt,
                                             PostParams pp, float* __restrict__ xt_out, float* __restrict__ prob_out,
                                             long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  xt_out[idx] = categorical_step(logits[idx * 2], logits[idx * 2 + 1], xt[idx], pp, idx,
                                 prob_out ? prob_out + idx : nullptr);
}

__global__ void gaussian_posterior_kernel(const float* __restrict__ pred, const float* __restrict__ xt, PostParams pp,
                                          float* __restrict__ xt_out, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  xt_out[idx] = gaussian_step(pred[idx], xt[idx], pp, idx);
}

// ------------------------------------------------------------------------------------------------
// Tiled-layout (edge_tiled_offset, H = 256) variants used by the fused path: every wave instruction moves
// one contiguous KiB.
// ------------------------------------------------------------------------------------------------
// e0 rows from the 2-row embedding table.  Thread = one float4 of the tiled buffer; positions of pad
// edges (s >= rows) are left untouched (the driver zeroes the tail once per step).
__global__ __launch_bounds__(256) void table_rows_tiled_kernel(const float* __restrict__ x, const int* __restrict__ perm,
                                                               const float* __restrict__ table, long long rows,
                                                               float* __restrict__ out) {
  const long long p4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index
  const long long tile = p4 >> 11;                                             // 2048 float4 per tile
  const int r = (int)(p4 & 2047), ks = r >> 7, i = (r >> 6) & 1, lane = r & 63;
  const long long s = tile * 32 + (lane & 31);
  if (s >= rows) return;
  const int f = 16 * ks + 8 * i + 4 * (lane >> 5);
  const float xv = x[perm ? perm[s] : s];
  *reinterpret_cast<v4f*>(out + p4 * 4) = *reinterpret_cast<const v4f*>(table + (xv > 0.5f ? 256 : 0) + f);
}

// max |e| per 32-edge tile of the tiled buffer (the e-stream scale of the fused kernel's fp16 planes) for producers of e
// that do not emit it themselves.  One wavefront per tile: 32 float4 per lane.
__global__ __launch_bounds__(256) void tile_absmax_tiled_kernel(const float* __restrict__ feat, long long n_tiles,
                                                                float* __restrict__ tile_max) {
  const int lane = threadIdx.x & 63;
  const long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= n_tiles) return;
  float m = 0.0f;
#pragma unroll 8
  for (int c = 0; c < 32; ++c) {
    const v4f x = *reinterpret_cast<const v4f*>(feat + t * 8192 + c * 256 + lane * 4);
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
  if (lane == 0) tile_max[t] = m;
}

// GroupNorm partial sums on the tiled buffer: a KiB chunk (tile, c = 2 ks + i) holds 8 channels = ONE group c of
// 32 edges.  Block b: groups 4 (b % 8) + wave, tiles b / 8, b / 8 + gridDim.x / 8, ...  Pad lanes hold zeros.
__global__ __launch_bounds__(256) void gn_partial_tiled_kernel(const float* __restrict__ feat, long long n_tiles,
                                                               double* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = 4 * (blockIdx.x & 7) + wave;
  double s = 0.0, q = 0.0;
  for (long long t = blockIdx.x >> 3; t < n_tiles; t += gridDim.x >> 3) {
    const v4f x = *reinterpret_cast<const v4f*>(feat + t * 8192 + g * 256 + lane * 4);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      s += (double)x[v];
      q += (double)x[v] * (double)x[v];
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (lane == 0) {
    partial[(long long)blockIdx.x * 64 + g * 2 + 0] = s;
    partial[(long long)blockIdx.x * 64 + g * 2 + 1] = q;
  }
}

// The same per STATISTIC SEGMENT (dense mode: one segment per sample, gnn_encoder.py:380; the segments need not be tile
// aligned: rows outside [seg_ptr[s], seg_ptr[s+1]) of a straddling tile are masked).  grid (blocks, segments); wave w of a
// block owns the groups 8 w .. 8 w + 7 of the tiles t0 + blockIdx.x, + gridDim.x, ...; every block writes all 32 groups
// (partial[(seg * gridDim.x + block) * 64 + 2 g + {0, 1}]: the all-groups-per-block layout of gn_finalize_kernel).
__global__ __launch_bounds__(256) void gn_partial_tiled_seg_kernel(const float* __restrict__ feat, const int* __restrict__ seg_ptr,
                                                                   double* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
  const int seg = blockIdx.y;
  const long long r0 = seg_ptr[seg], r1 = seg_ptr[seg + 1];
  double s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.0;
  if (r1 > r0) {
    const long long t0 = r0 >> 5, t1 = (r1 - 1) >> 5;
    for (long long t = t0 + blockIdx.x; t <= t1; t += gridDim.x) {
      const long long row = t * 32 + l31;
      if (row < r0 || row >= r1) continue;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const v4f x = *reinterpret_cast<const v4f*>(feat + t * 8192 + (8 * wave + k) * 256 + lane * 4);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          s[k] += (double)x[v];
          q[k] += (double)x[v] * (double)x[v];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s[k] += __shfl_xor(s[k], off, 64);
      q[k] += __shfl_xor(q[k], off, 64);
    }
  }
  if (lane == 0) {
    double* dst = partial + ((long long)seg * gridDim.x + blockIdx.x) * 64 + 16 * wave;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dst[2 * k] = s[k];
      dst[2 * k + 1] = q[k];
    }
  }
}

// GroupNorm partial sums that the last fused edge layer left per tile (gn_tile[tile][32 groups][sum, sumsq] floats)
// -> partial[256 blocks][64] doubles in the all-groups-per-block layout of gn_finalize_kernel (group stride 1).
__global__ __launch_bounds__(256) void gn_tiles_reduce_kernel(const float* __restrict__ gn_tile, long long n_tiles,
                                                              double* __restrict__ partial) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  double acc = 0.0;
  for (long long t = (long long)blockIdx.x * 4 + r; t < n_tiles; t += (long long)gridDim.x * 4) acc += (double)gn_tile[t * 64 + c];
  red[r][c] = acc;
  __syncthreads();
  if (r == 0) partial[(long long)blockIdx.x * 64 + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// head on the tiled buffer: one wavefront per 32-edge tile; lane (l31, hh) accumulates the conv dot products of
// edge l31 over its half of the channels, one cross-half exchange, then 32 lanes finish 32 edges at once.
// SEG: per-row statistic segments (seg_ptr [n_segments + 1], stats [n_segments][32][2]): every lane looks its edge's
// segment up (binary search) and reads that segment's mean / rstd from memory instead of the block's LDS copy.
template <int C, bool SEG>
__global__ __launch_bounds__(256) void head_apply_tiled_kernel(const float* __restrict__ feat, long long rows,
                                                               long long n_tiles, const float* __restrict__ stats,
                                                               const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                               const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                                               const int* __restrict__ perm, const float* __restrict__ xt,
                                                               PostParams pp, float* __restrict__ xt_out,
                                                               float* __restrict__ pred_out, float* __restrict__ prob_out,
                                                               const int* __restrict__ seg_ptr, int n_segments) {
  // statistics and the per-channel parameters sit in LDS: every tile re-reads them, 4 channels per lane and chunk
  __shared__ float st[64];
  __shared__ __attribute__((aligned(16))) float s_gw[256], s_gb[256], s_cw[C][256];
  if (threadIdx.x < 64) st[threadIdx.x] = stats[threadIdx.x];
  s_gw[threadIdx.x] = gn_w[threadIdx.x];
  s_gb[threadIdx.x] = gn_b[threadIdx.x];
#pragma unroll
  for (int c = 0; c < C; ++c) s_cw[c][threadIdx.x] = conv_w[c * 256 + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long t = wave0; t < n_tiles; t += nwaves) {
    const float* tp = feat + t * 8192 + lane * 4;
    const float* st_row = stats;
    if constexpr (SEG) {      // last segment whose first row is <= this lane's edge (pad lanes: the last segment)
      long long srow = t * 32 + l31;
      srow = srow < rows ? srow : rows - 1;
      int lo = 0, hi = n_segments - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((long long)seg_ptr[mid] <= srow) lo = mid; else hi = mid - 1;
      }
      st_row = stats + (long long)lo * 64;
    }
    float dot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dot[c] = 0.0f;
#pragma unroll 8
    for (int ch = 0; ch < 32; ++ch) {                 // chunk ch = 2 ks + i = GroupNorm group
      // (last use of e in the step: non-temporal - the streaming read runs ~10 % faster that way, profiles/r04/reread_probe.txt)
      const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(tp + ch * 256));
      const int f = 8 * ch + 4 * hh;
      const float mean = SEG ? st_row[2 * ch] : st[2 * ch], rstd = SEG ? st_row[2 * ch + 1] : st[2 * ch + 1];
      const v4f gw = *reinterpret_cast<const v4f*>(s_gw + f), gb = *reinterpret_cast<const v4f*>(s_gb + f);
      v4f cw[C];
#pragma unroll
      for (int c = 0; c < C; ++c) cw[c] = *reinterpret_cast<const v4f*>(&s_cw[c][f]);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float y = (x[v] - mean) * rstd * gw[v] + gb[v];
        y = y > 0.0f ? y : 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) dot[c] += y * cw[c][v];
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dot[c] += __shfl_xor(dot[c], 32, 64);
    const long long s = t * 32 + l31;
    if (hh == 0 && s < rows) {
      const long long idx = perm ? perm[s] : s;
      if constexpr (C == 2) {
        const float l0 = dot[0] + conv_b[0], l1 = dot[1] + conv_b[1];
        if (pred_out) {
          pred_out[idx * 2 + 0] = l0;
          pred_out[idx * 2 + 1] = l1;
        }
        xt_out[idx] = categorical_step(l0, l1, xt[idx], pp, idx, prob_out ? prob_out + idx : nullptr);
      } else {
        const float pred = dot[0] + conv_b[0];
        if (pred_out) pred_out[idx] = pred;
        xt_out[idx] = gaussian_step(pred, xt[idx], pp, idx);
      }
    }
  }
}

// ================================================================================================
// host-side launchers
// ================================================================================================
#define DIFUSCO_VEC_DISPATCH(H, CALL)      \
  switch (H) {                             \
    case 64: { constexpr int VEC = 1; CALL; } break;  \
    case 128: { constexpr int VEC = 2; CALL; } break; \
    case 256: { constexpr int VEC = 4; CALL; } break; \
    default: return hipErrorInvalidValue;  \
  }

hipError_t launch_time_bias(const float* t_host, int n_t, int H, int n_layers, const float* freqs, const float* w0,
                            const float* b0, const float* w2, const float* b2, const float* wl_base, long long layer_stride,
                            long long wl_w_off, long long wl_b_off, float* tbias, hipStream_t stream) {
  if (H > 256 || n_t < 1) return hipErrorInvalidValue;
  for (int first = 0; first < n_t; first += 64) {      // the times travel as kernel arguments: no device copy, capture safe
    TimeBatch tb;
    const int n = n_t - first < 64 ? n_t - first : 64;
    for (int i = 0; i < 64; ++i) tb.t[i] = t_host[first + (i < n ? i : 0)];
    hipLaunchKernelGGL(time_bias_kernel, dim3(n_layers, n), dim3(256), 0, stream, tb, H, freqs, w0, b0, w2, b2, wl_base,
                       layer_stride, wl_w_off, wl_b_off, tbias + (long long)first * n_layers * H);
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) return er;
  }
  return hipSuccess;
}

hipError_t launch_pos_embed(const float* points, const float* dimt, int n_nodes, int H, float* out, hipStream_t stream) {
  const long long n = (long long)n_nodes * H;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pos_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, points, dimt, n_nodes, H, out);
  return hipGetLastError();
}

hipError_t launch_scalar_embed(const float* x, const int* perm, const float* dimt, long long rows, int H, float* out,
                               hipStream_t stream) {
  const long long n = rows * H;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scalar_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, perm, dimt, rows, H, out);
  return hipGetLastError();
}

hipError_t launch_table_rows(const float* x, const int* perm, const float* table, long long rows, int H, float* out,
                             hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  long long blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((table_rows_kernel<VEC>), dim3((unsigned)blocks), dim3(256), 0, stream, x,
                                             perm, table, rows, out))
  return hipGetLastError();
}

hipError_t launch_edge_gate_aggregate(int H, int n_nodes, const int* rowptr, const int* col, const float* node4,
                                      float* ce_act, float* h, const float* nh_w, const float* nh_b, const float* ne_w,
                                      const float* ne_b, const float* ol_w, const float* ol_b, const float* tbias,
                                      int time_on_edge, hipStream_t stream, int agg_mode) {
  if (n_nodes == 0) return hipSuccess;
  if (agg_mode < 0 || agg_mode > 2) return hipErrorInvalidValue;
  const unsigned blocks = (unsigned)((n_nodes + 3) / 4);
  DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((edge_gate_aggregate_kernel<VEC>), dim3(blocks), dim3(256), 0, stream,
                                             n_nodes, rowptr, col, node4, ce_act, h, nh_w, nh_b, ne_w, ne_b, ol_w, ol_b,
                                             tbias, time_on_edge, agg_mode))
  return hipGetLastError();
}

int gn_blocks_for(long long rows) {
  long long b = (rows + 63) / 64;  // >= 16 rows per wave
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

hipError_t launch_head(int H, int C, const float* feat, const int* seg_ptr, int n_segments, long long total_rows,
                       int nblk, double* partial, float* stats, const float* gn_w, const float* gn_b,
                       const float* conv_w, const float* conv_b, const int* perm, const float* xt, const float* post,
                       int rand_mode, const float* rand, unsigned long long seed, unsigned long long offset,
                       float* xt_out, float* pred_out, float* prob_out, hipStream_t stream, int gn_phase,
                       double* gn_sums) {
  if (total_rows == 0) return hipSuccess;
  if (gn_phase != 0 && (n_segments != 1 || !gn_sums)) return hipErrorInvalidValue;
  PostParams pp;
  for (int i = 0; i < 8; ++i) pp.p[i] = post[i];
  pp.rand_mode = rand_mode;
  pp.rand = rand;
  pp.seed = seed;
  pp.offset = offset;
  dim3 grid((unsigned)nblk, (unsigned)n_segments);
  if (gn_phase != 2) {
    DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((gn_partial_kernel<VEC>), grid, dim3(256), 0, stream, feat, seg_ptr,
                                               total_rows, partial))
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_segments), dim3(1024), 0, stream, partial, seg_ptr, total_rows, nblk, H / 32,
                       1, stats, gn_phase == 1 ? gn_sums : (double*)nullptr);
    if (gn_phase == 1) return hipGetLastError();
  } else {
    hipLaunchKernelGGL(gn_stats_from_sums_kernel, dim3(1), dim3(64), 0, stream, gn_sums, H / 32, stats);
  }
  if (C == 2) {
    DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((head_apply_kernel<VEC, 2>), grid, dim3(256), 0, stream, feat, seg_ptr,
                                               total_rows, stats, gn_w, gn_b, conv_w, conv_b, perm, xt, pp, xt_out,
                                               pred_out, prob_out))
  } else if (C == 1) {
    DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((head_apply_kernel<VEC, 1>), grid, dim3(256), 0, stream, feat, seg_ptr,
                                               total_rows, stats, gn_w, gn_b, conv_w, conv_b, perm, xt, pp, xt_out,
                                               pred_out, prob_out))
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_categorical_posterior(const float* logits, const float* xt, const float* post, int rand_mode,
                                        const float* rand, unsigned long long seed, unsigned long long offset,
                                        float* xt_out, float* prob_out, long long n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  PostParams pp;
  for (int i = 0; i < 8; ++i) pp.p[i] = post[i];
  pp.rand_mode = rand_mode; pp.rand = rand; pp.seed = seed; pp.offset = offset;
  hipLaunchKernelGGL(categorical_posterior_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, logits, xt, pp,
                     xt_out, prob_out, n);
  return hipGetLastError();
}

hipError_t launch_gaussian_posterior(const float* pred, const float* xt, const float* post, int rand_mode,
                                     const float* rand, unsigned long long seed, unsigned long long offset, float* xt_out,
                                     long long n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  PostParams pp;
  for (int i = 0; i < 8; ++i) pp.p[i] = post[i];
  pp.rand_mode = rand_mode; pp.rand = rand; pp.seed = seed; pp.offset = offset;
  hipLaunchKernelGGL(gaussian_posterior_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pred, xt, pp,
                     xt_out, n);
  return hipGetLastError();
}

__global__ void count_nonfinite_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ count) {
  unsigned bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    bad += !__builtin_isfinite(x[i]);
  if (bad) atomicAdd(count, bad);
}

hipError_t launch_count_nonfinite(const float* x, long long n, unsigned* count, hipStream_t stream) {
  if (n <= 0 || x == nullptr) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, count);
  return hipGetLastError();
}

hipError_t launch_tile_absmax_tiled(const float* e, long long n_tiles, float* tile_max, hipStream_t stream) {
  if (n_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(tile_absmax_tiled_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, stream, e, n_tiles, tile_max);
  return hipGetLastError();
}

hipError_t launch_table_rows_tiled(const float* x, const int* perm, const float* table, long long rows, float* out,
                                   hipStream_t stream) {
  if (rows == 0) return hipSuccess;
  const long long n4 = ((rows + 31) / 32) * 2048;     // float4 slots of all tiles that hold at least one edge
  hipLaunchKernelGGL(table_rows_tiled_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, perm, table, rows, out);
  return hipGetLastError();
}

// y[r][f] = b[f] + sum_k W[f][k] x[r][k] for r = 0, 1 (the two rows of the layer-0 tables): one wavefront per output
// feature, lanes over k, fp32 fma chain per lane + wave reduction.  A [2 x H] problem is far below what the MFMA row
// kernels are built for (their fixed cost is ~50 us); this takes a few microseconds.
template <int VEC>
__global__ __launch_bounds__(256) void two_rows_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y) {
  constexpr int H = 64 * VEC;
  const int lane = threadIdx.x & 63, f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= H) return;
  float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int k = lane * VEC + v;
    const float wv = w[(long long)f * H + k];
    a0 = fmaf(wv, x[k], a0);
    a1 = fmaf(wv, x[H + k], a1);
  }
  wave_sum2(a0, a1);
  if (lane == 0) {
    const float b = bias ? bias[f] : 0.0f;
    y[f] = a0 + b;
    y[H + f] = a1 + b;
  }
}

hipError_t launch_two_rows_linear(int H, const float* x, const float* w, const float* bias, float* y, hipStream_t stream) {
  dim3 grid((unsigned)(H / 4));
  DIFUSCO_VEC_DISPATCH(H, hipLaunchKernelGGL((two_rows_linear_kernel<VEC>), grid, dim3(256), 0, stream, x, w, bias, y))
  return hipGetLastError();
}

// GroupNorm + conv + posterior on the tiled e buffer (one statistic segment = the whole call).
// partial must hold nblk*64 doubles, nblk a multiple of 8.
hipError_t launch_head_tiled(int C, const float* feat, long long rows, int nblk, double* partial, float* stats,
                             const float* gn_w, const float* gn_b, const float* conv_w, const float* conv_b,
                             const int* perm, const float* xt, const float* post, int rand_mode, const float* rand,
                             unsigned long long seed, unsigned long long offset, float* xt_out, float* pred_out,
                             float* prob_out, hipStream_t stream, const float* gn_tile, int gn_phase, double* gn_sums,
                             const int* seg_ptr, int n_segments) {
  if (rows == 0) return hipSuccess;
  if (gn_phase != 0 && !gn_sums) return hipErrorInvalidValue;
  if (n_segments > 1) {      // per-segment statistics (dense mode: one segment per sample): a masked pass over e per segment
    if (!seg_ptr || gn_phase != 0) return hipErrorInvalidValue;
    PostParams pq;
    for (int i = 0; i < 8; ++i) pq.p[i] = post[i];
    pq.rand_mode = rand_mode; pq.rand = rand; pq.seed = seed; pq.offset = offset;
    const long long n_tiles_s = (rows + 31) / 32;
    int bps = (int)((n_tiles_s / n_segments + 3) / 4);      // ~4 tiles per block
    bps = bps < 1 ? 1 : (bps > 256 ? 256 : bps);
    hipLaunchKernelGGL(gn_partial_tiled_seg_kernel, dim3(bps, n_segments), dim3(256), 0, stream, feat, seg_ptr, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_segments), dim3(1024), 0, stream, partial, seg_ptr, rows, bps, 8, 1, stats,
                       (double*)nullptr);
    long long blocks_s = (n_tiles_s + 3) / 4;
    if (blocks_s > 4096) blocks_s = 4096;
    if (C == 2) {
      hipLaunchKernelGGL((head_apply_tiled_kernel<2, true>), dim3((unsigned)blocks_s), dim3(256), 0, stream, feat, rows, n_tiles_s,
                         stats, gn_w, gn_b, conv_w, conv_b, perm, xt, pq, xt_out, pred_out, prob_out, seg_ptr, n_segments);
    } else if (C == 1) {
      hipLaunchKernelGGL((head_apply_tiled_kernel<1, true>), dim3((unsigned)blocks_s), dim3(256), 0, stream, feat, rows, n_tiles_s,
                         stats, gn_w, gn_b, conv_w, conv_b, perm, xt, pq, xt_out, pred_out, prob_out, seg_ptr, n_segments);
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  if (nblk < 8 || nblk % 8 != 0) return hipErrorInvalidValue;   // (with gn_tile, partial must hold 256 * 64 doubles)
  PostParams pp;
  for (int i = 0; i < 8; ++i) pp.p[i] = post[i];
  pp.rand_mode = rand_mode; pp.rand = rand; pp.seed = seed; pp.offset = offset;
  const long long n_tiles = (rows + 31) / 32;
  double* sums_out = gn_phase == 1 ? gn_sums : (double*)nullptr;
  if (gn_phase == 2) {
    hipLaunchKernelGGL(gn_stats_from_sums_kernel, dim3(1), dim3(64), 0, stream, gn_sums, 8, stats);
  } else if (gn_tile) {   // statistics come from the last fused layer: no pass over feat
    hipLaunchKernelGGL(gn_tiles_reduce_kernel, dim3(256), dim3(256), 0, stream, gn_tile, n_tiles, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(1), dim3(1024), 0, stream, partial, (const int*)nullptr, rows, 256, 8, 1, stats,
                       sums_out);
  } else {
    hipLaunchKernelGGL(gn_partial_tiled_kernel, dim3(nblk), dim3(256), 0, stream, feat, n_tiles, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(1), dim3(1024), 0, stream, partial, (const int*)nullptr, rows, nblk, 8, 8, stats,
                       sums_out);
  }
  if (gn_phase == 1) return hipGetLastError();
  long long blocks = (n_tiles + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (C == 2) {
    hipLaunchKernelGGL((head_apply_tiled_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, stream, feat, rows, n_tiles, stats,
                       gn_w, gn_b, conv_w, conv_b, perm, xt, pp, xt_out, pred_out, prob_out, (const int*)nullptr, 1);
  } else if (C == 1) {
    hipLaunchKernelGGL((head_apply_tiled_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, stream, feat, rows, n_tiles, stats,
                       gn_w, gn_b, conv_w, conv_b, perm, xt, pp, xt_out, pred_out, prob_out, (const int*)nullptr, 1);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace difusco
