// MCTS heatmap rows on the GPU: the numeric part of tsp_mcts/convert_numpy_to_txt.py:18-47 (reader:
// tsp_mcts/code/include/TSP_IO.h:461-492) from the SPARSE E-entry heatmap, float32, nothing N x N in memory.
//
//   v[i][j] = heat[i][j] + 0.01 * (1 - |p_i - p_j|)          (:21-22; the distance prior is dense, heat is not)
//   T       = the k-th largest positive v, k = int(N * N * prob)                                   (:26-29)
//   keep    = { v > T }  U  { the 3 largest entries of every row }                                   (:33-43)
//   m       = v * keep, + 0.01 on its non-zeros;  out = (m + m^T) / rowsum(m + m^T)                  (:44-47)
//
// Every one of the N^2 values is a candidate (at N = 10^4 the threshold lies among entries that carry the prior alone)
// and the output is N^2 numbers, so the work is O(N^2); it is recomputed from the E entries wherever it is needed:
//   prepare   sort the entries by (row, col) and by (col, row) (rocPRIM) -> CSR of heat and of its transpose;
//             pass 1 (one workgroup per row): radix histogram of the float32 bit patterns of the positive values
//             (positive floats order like their bits), level 1 = bits 31..21, and the row top-3;
//             passes 2, 3: levels 20..10 and 9..0 inside the bucket that holds the k-th largest -> T exactly;
//   block     one workgroup per output row: both orientations of every pair, masks, the row in LDS, the row sum in
//             numpy's own order, the division.
// Bit-exactness with the numpy program (the text is %.6f of these numbers, and 10^8 of them are printed): no fused
// multiply-add anywhere (numpy has none), correctly rounded sqrt / divide, the converter's operation order, and the row
// sum as numpy's float32 add.reduce computes it: chunks of 8192 elements (the ufunc buffer), each by pairwise summation -
// blocks of <= 128 with eight interleaved accumulators, split points rounded down to a multiple of 8 - added to a running
// total starting at 0.  tests/test_host_logic.py checks that summation tree against numpy itself.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/difusco_hip.h"
#include "kernels.h"

namespace difusco {
namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 8192;       // numpy's ufunc buffer size in elements (np.getbufsize())

struct Header {                    // first 256 bytes of the workspace
  int n, prepared;
  long long n_edges;
  float threshold;
  int n_leaves, n_ops;
  long long k_requested, positives;
};

struct Carve {
  Header* hdr;
  unsigned long long *key_a, *key_b, *hist;
  unsigned *val_a, *val_b;
  int *rowptr, *col, *rowptr_t, *col_t, *top3, *leaves, *ops, *flags;
  float *heat, *heat_t;
  void* temp;
  size_t temp_bytes, total;
};

size_t up256(size_t x) { return (x + 255) / 256 * 256; }

// numpy's pairwise summation tree over n elements, as leaves (lo, len) and a combine program.  Slots: leaf t -> slot t;
// op (dst, a, b): slot[dst] = slot[a] + slot[b]; the LAST op's dst holds the row sum.  Chunks of kChunk are added to a
// running total that starts at +0.0 (0 + x is exact).
void build_sum_program(int n, std::vector<int>& leaves, std::vector<int>& ops) {
  leaves.clear();
  ops.clear();
  int next_slot = 0;
  std::vector<int> leaf_slots;
  // first pass: enumerate leaves in order so that leaf t uses slot t
  struct Rec {
    static void leaves_of(int lo, int len, std::vector<int>& out) {
      if (len <= 128) {
        out.push_back(lo);
        out.push_back(len);
        return;
      }
      int n2 = len / 2;
      n2 -= n2 % 8;
      leaves_of(lo, n2, out);
      leaves_of(lo + n2, len - n2, out);
    }
  };
  for (int lo = 0; lo < n; lo += kChunk) Rec::leaves_of(lo, (n - lo < kChunk ? n - lo : kChunk), leaves);
  const int n_leaves = (int)leaves.size() / 2;
  next_slot = n_leaves;
  int leaf_cursor = 0;
  // second pass: the same recursion, emitting combine ops (post-order)
  struct Emit {
    static int run(int len, int& leaf_cursor, int& next_slot, std::vector<int>& ops) {
      if (len <= 128) return leaf_cursor++;
      int n2 = len / 2;
      n2 -= n2 % 8;
      const int a = run(n2, leaf_cursor, next_slot, ops);
      const int b = run(len - n2, leaf_cursor, next_slot, ops);
      const int d = next_slot++;
      ops.push_back(d);
      ops.push_back(a);
      ops.push_back(b);
      return d;
    }
  };
  int total = -1;                                  // slot of the running total (-1: the initial +0.0)
  for (int lo = 0; lo < n; lo += kChunk) {
    const int c = Emit::run((n - lo < kChunk ? n - lo : kChunk), leaf_cursor, next_slot, ops);
    const int d = next_slot++;
    ops.push_back(d);
    ops.push_back(total);                          // -1 = 0.0f
    ops.push_back(c);
    total = d;
  }
}

hipError_t carve(void* base, int n, long long E, Carve* c) {
  size_t t1 = 0;
  hipError_t er = rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                            (unsigned*)nullptr, (unsigned*)nullptr, (size_t)(E > 0 ? E : 1), 0, 64, 0, false);
  if (er != hipSuccess) return er;
  c->temp_bytes = t1;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    size_t at = cur;
    cur += up256(bytes);
    return base ? (void*)((char*)base + at) : (void*)nullptr;
  };
  const size_t Ez = (size_t)(E > 0 ? E : 1);
  c->hdr = (Header*)take(256);
  c->key_a = (unsigned long long*)take(8 * Ez);
  c->key_b = (unsigned long long*)take(8 * Ez);
  c->val_a = (unsigned*)take(4 * Ez);
  c->val_b = (unsigned*)take(4 * Ez);
  c->hist = (unsigned long long*)take(8 * 2048);
  c->rowptr = (int*)take(4 * ((size_t)n + 1));
  c->col = (int*)take(4 * Ez);
  c->heat = (float*)take(4 * Ez);
  c->rowptr_t = (int*)take(4 * ((size_t)n + 1));
  c->col_t = (int*)take(4 * Ez);
  c->heat_t = (float*)take(4 * Ez);
  c->top3 = (int*)take(4 * 3 * (size_t)n);
  c->leaves = (int*)take(4 * 2 * ((size_t)n / 8 + 64));
  c->ops = (int*)take(4 * 3 * ((size_t)n / 8 + 64));
  c->flags = (int*)take(256);
  c->temp = take(t1);
  c->total = cur;
  return hipSuccess;
}

__global__ void key_kernel(const int* __restrict__ row, const int* __restrict__ col, long long E, long long n, int transposed,
                           unsigned long long* __restrict__ key, unsigned* __restrict__ val, int* __restrict__ flags) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const long long i = row[e], j = col[e];
  if (i < 0 || i >= n || j < 0 || j >= n) {
    atomicOr(flags, 1);
    key[e] = ~0ull;
  } else {
    key[e] = (unsigned long long)(transposed ? j * n + i : i * n + j);
  }
  val[e] = (unsigned)e;
}

// sorted keys -> CSR arrays: rowptr by binary search (one thread per row), minor index + heat by gather
__global__ void csr_kernel(const unsigned long long* __restrict__ key, const unsigned* __restrict__ val,
                           const float* __restrict__ heat_in, long long E, long long n, int* __restrict__ rowptr,
                           int* __restrict__ minor, float* __restrict__ heat, int* __restrict__ flags) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= n) {
    const unsigned long long want = (unsigned long long)t * (unsigned long long)n;
    long long lo = 0, hi = E;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (key[mid] < want) lo = mid + 1; else hi = mid;
    }
    rowptr[t] = (int)lo;
  }
  if (t < E) {
    minor[t] = (int)(key[t] % (unsigned long long)n);
    heat[t] = heat_in[val[t]];
    if (t > 0 && key[t] == key[t - 1]) atomicOr(flags, 2);          // duplicate (row, col) entry
  }
}

// the converter's value of the ordered pair (a, b): heat_ab + 0.01 * (1 - |p_a - p_b|), every operation rounded to float32
// like numpy evaluates it (norm = sqrt(dx*dx + dy*dy), no fused multiply-add), +inf -> 0 (:24)
__device__ __forceinline__ float prior_value(float heat_ab, float ax, float ay, float bx, float by) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by);
  // (sqrtf, not __fsqrt_rn: HIP maps the latter to the NATIVE, approximate square root; this translation unit is compiled
  // with -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off, build.py)
  const float d = __builtin_sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
  const float v = __fadd_rn(heat_ab, __fmul_rn(0.01f, __fsub_rn(1.0f, d)));
  return v == __builtin_inff() ? 0.0f : v;
}

// sparse row lookup for a thread that visits j in increasing order: entry (row, j) of a CSR row, 0 if absent
struct RowCursor {
  const int* minor;
  const float* heat;
  int cur, end;
  __device__ float at(int j) {
    while (cur < end && minor[cur] < j) ++cur;
    return (cur < end && minor[cur] == j) ? heat[cur] : 0.0f;
  }
};

// larger of two (value, index) candidates in ascending-argsort order: by value, equal values by index (the later one
// in a stable ascending sort is the one with the larger index)
__device__ __forceinline__ bool ranks_above(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia > ib); }

__device__ __forceinline__ void top3_insert(float (&tv)[3], int (&ti)[3], float v, int j) {
  if (ti[2] >= 0 && !ranks_above(v, j, tv[2], ti[2])) return;
  tv[2] = v; ti[2] = j;
  if (ti[1] < 0 || ranks_above(tv[2], ti[2], tv[1], ti[1])) {
    float fv = tv[1]; int fi = ti[1]; tv[1] = tv[2]; ti[1] = ti[2]; tv[2] = fv; ti[2] = fi;
    if (ti[0] < 0 || ranks_above(tv[1], ti[1], tv[0], ti[0])) {
      fv = tv[0]; fi = ti[0]; tv[0] = tv[1]; ti[0] = ti[1]; tv[1] = fv; ti[1] = fi;
    }
  }
}

// LEVEL 1: histogram of bits 31..21 of every positive v of row blockIdx.x, and the row's top 3 (any sign).
// LEVEL 2 / 3: bits 20..10 / 9..0 of the positive values whose higher bits equal `prefix`.
template <int LEVEL>
__global__ __launch_bounds__(kThreads) void select_pass_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                               const float* __restrict__ heat, const float* __restrict__ pts,
                                                               unsigned prefix, unsigned long long* __restrict__ hist,
                                                               int* __restrict__ top3) {
  __shared__ unsigned lh[2048];
  __shared__ float s_tv[kThreads][3];
  __shared__ int s_ti[kThreads][3];
  const int i = blockIdx.x, tid = threadIdx.x;
  for (int b = tid; b < 2048; b += kThreads) lh[b] = 0;
  __syncthreads();
  const float ax = pts[2 * i], ay = pts[2 * i + 1];
  RowCursor rc{col, heat, rowptr[i], rowptr[i + 1]};
  float tv[3] = {0.0f, 0.0f, 0.0f};
  int ti[3] = {-1, -1, -1};
  for (int j = tid; j < n; j += kThreads) {
    const float v = prior_value(rc.at(j), ax, ay, pts[2 * j], pts[2 * j + 1]);
    if (LEVEL == 1) top3_insert(tv, ti, v, j);
    if (v > 0.0f) {
      const unsigned bits = __builtin_bit_cast(unsigned, v);
      if (LEVEL == 1) atomicAdd(&lh[bits >> 21], 1u);
      else if (LEVEL == 2) { if ((bits >> 21) == prefix) atomicAdd(&lh[(bits >> 10) & 2047u], 1u); }
      else { if ((bits >> 10) == prefix) atomicAdd(&lh[bits & 1023u], 1u); }
    }
  }
  __syncthreads();
  for (int b = tid; b < 2048; b += kThreads)
    if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
  if (LEVEL == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { s_tv[tid][k] = tv[k]; s_ti[tid][k] = ti[k]; }
    __syncthreads();
    if (tid == 0) {
      float bv[3] = {0.0f, 0.0f, 0.0f};
      int bi[3] = {-1, -1, -1};
      for (int t = 0; t < kThreads; ++t)
        for (int k = 0; k < 3; ++k)
          if (s_ti[t][k] >= 0) top3_insert(bv, bi, s_tv[t][k], s_ti[t][k]);
      for (int k = 0; k < 3; ++k) top3[3 * i + k] = bi[k];       // (rows shorter than 3: -1, matches nothing)
    }
  }
}

// one workgroup per output row i = row_begin + blockIdx.x; dynamic LDS: n floats (the row) + slot array of the sum program
__global__ __launch_bounds__(kThreads) void rows_kernel(int n, int row_begin, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const float* __restrict__ heat, const int* __restrict__ rowptr_t,
                                                        const int* __restrict__ col_t, const float* __restrict__ heat_t,
                                                        const float* __restrict__ pts, const int* __restrict__ top3, float thr,
                                                        const int* __restrict__ leaves, int n_leaves, const int* __restrict__ ops,
                                                        int n_ops, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rowv = lds;                 // [n]
  float* slot = lds + n;             // [n_leaves + n_ops]
  const int i = row_begin + blockIdx.x, tid = threadIdx.x;
  const float ax = pts[2 * i], ay = pts[2 * i + 1];
  const int t0 = top3[3 * i], t1 = top3[3 * i + 1], t2 = top3[3 * i + 2];
  RowCursor fwd{col, heat, rowptr[i], rowptr[i + 1]};          // heat[i][j]
  RowCursor bwd{col_t, heat_t, rowptr_t[i], rowptr_t[i + 1]};  // heat[j][i]
  for (int j = tid; j < n; j += kThreads) {
    const float bx = pts[2 * j], by = pts[2 * j + 1];
    const float v = prior_value(fwd.at(j), ax, ay, bx, by);                    // v[i][j]
    const float w = prior_value(bwd.at(j), bx, by, ax, ay);                    // v[j][i] (its own operand order: p_j - p_i)
    const bool kv = v > thr || j == t0 || j == t1 || j == t2;                  // :35, :41-43
    const bool kw = w > thr || i == top3[3 * j] || i == top3[3 * j + 1] || i == top3[3 * j + 2];
    float m = __fmul_rn(v, kv ? 1.0f : 0.0f);                                  // :44  (negative * False = -0.0)
    float mt = __fmul_rn(w, kw ? 1.0f : 0.0f);
    if (m != 0.0f) m = __fadd_rn(m, 0.01f);                                    // :45
    if (mt != 0.0f) mt = __fadd_rn(mt, 0.01f);
    rowv[j] = __fadd_rn(m, mt);                                                // :46
  }
  __syncthreads();
  // :47 row sum in numpy's order
  for (int t = tid; t < n_leaves; t += kThreads) {
    const float* a = rowv + leaves[2 * t];
    const int len = leaves[2 * t + 1];
    float res;
    if (len < 8) {
      res = 0.0f;
      for (int k = 0; k < len; ++k) res = __fadd_rn(res, a[k]);
    } else {
      float r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = a[k];
      int p = 8;
      for (; p < len - (len % 8); p += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __fadd_rn(r[k], a[p + k]);
      }
      res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
      for (; p < len; ++p) res = __fadd_rn(res, a[p]);
    }
    slot[t] = res;
  }
  __syncthreads();
  if (tid == 0) {
    for (int o = 0; o < n_ops; ++o) {
      const int d = ops[3 * o], a = ops[3 * o + 1], b = ops[3 * o + 2];
      slot[d] = __fadd_rn(a < 0 ? 0.0f : slot[a], slot[b]);
    }
  }
  __syncthreads();
  const float total = slot[ops[3 * (n_ops - 1)]];
  float* orow = out + (long long)blockIdx.x * n;
  for (int j = tid; j < n; j += kThreads) orow[j] = rowv[j] / total;      // (correctly rounded: see prior_value)
}

int fail_hip(const char* what, hipError_t er) { return set_error(DIFUSCO_EHIP, "%s: %s", what, hipGetErrorString(er)); }

}  // namespace
}  // namespace difusco

extern "C" {

int difusco_mcts_heatmap_workspace_bytes(int n_nodes, int64_t n_edges, size_t* bytes) {
  using namespace difusco;
  if (!bytes || n_nodes < 1 || n_edges < 0) return set_error(DIFUSCO_EINVAL, "mcts_heatmap_workspace_bytes: bad arguments");
  Carve c;
  hipError_t er = carve(nullptr, n_nodes, n_edges, &c);
  if (er != hipSuccess) return fail_hip("rocprim size query", er);
  *bytes = c.total;
  return DIFUSCO_OK;
}

int difusco_mcts_heatmap_prepare(int n_nodes, int64_t n_edges, const int32_t* row, const int32_t* col, const float* heat,
                                 const float* points, double expected_valid_prob, void* workspace, size_t workspace_bytes,
                                 float* threshold_out, void* stream) {
  using namespace difusco;
  if (n_nodes < 1 || n_edges < 0 || (n_edges > 0 && (!row || !col || !heat)) || !points || !workspace)
    return set_error(DIFUSCO_EINVAL, "mcts_heatmap_prepare: bad arguments");
  if (n_nodes > 38000) return set_error(DIFUSCO_EUNSUPPORTED, "mcts_heatmap: n_nodes > 38000 (the output row is kept in LDS)");
  Carve c;
  hipError_t er = carve(workspace, n_nodes, n_edges, &c);
  if (er != hipSuccess) return fail_hip("rocprim size query", er);
  if (c.total > workspace_bytes) return set_error(DIFUSCO_EWORKSPACE, "mcts_heatmap: workspace too small: %zu < %zu", workspace_bytes, c.total);
  hipStream_t st = (hipStream_t)stream;
  const long long E = n_edges, n = n_nodes;
  er = hipMemsetAsync(c.flags, 0, 256, st);
  if (er != hipSuccess) return fail_hip("memset", er);
  const unsigned ge = (unsigned)((E + 255) / 256), gn = (unsigned)(((E > n + 1 ? E : n + 1) + 255) / 256);
  for (int tr = 0; tr < 2; ++tr) {
    if (E > 0) {
      hipLaunchKernelGGL(key_kernel, dim3(ge), dim3(256), 0, st, row, col, E, n, tr, c.key_a, c.val_a, c.flags);
      size_t tb = c.temp_bytes;
      er = rocprim::radix_sort_pairs(c.temp, tb, c.key_a, c.key_b, c.val_a, c.val_b, (size_t)E, 0, 64, st, false);
      if (er != hipSuccess) return fail_hip("radix_sort_pairs", er);
    }
    hipLaunchKernelGGL(csr_kernel, dim3(gn), dim3(256), 0, st, c.key_b, c.val_b, heat, E, n, tr ? c.rowptr_t : c.rowptr,
                       tr ? c.col_t : c.col, tr ? c.heat_t : c.heat, c.flags);
  }
  int flags = 0;
  er = hipMemcpyAsync(&flags, c.flags, sizeof(int), hipMemcpyDeviceToHost, st);
  if (er == hipSuccess) er = hipStreamSynchronize(st);
  if (er != hipSuccess) return fail_hip("flags", er);
  if (flags & 1) return set_error(DIFUSCO_EINVAL, "mcts_heatmap: edge endpoint out of range");
  if (flags & 2) return set_error(DIFUSCO_EINVAL, "mcts_heatmap: duplicate (row, col) entries in the sparse heatmap");

  // ---- threshold: three-level radix select over the positive values; row top-3 in the first pass --------------
  const long long k_req = (long long)((double)n * (double)n * expected_valid_prob);     // int(N * N * prob), :26
  std::vector<unsigned long long> h(2048);
  unsigned prefix = 0;
  long long rank = 0, positives = 0;                     // rank-th LARGEST inside the current prefix
  for (int level = 1; level <= 3; ++level) {
    er = hipMemsetAsync(c.hist, 0, 8 * 2048, st);
    if (er != hipSuccess) return fail_hip("memset", er);
    if (level == 1) hipLaunchKernelGGL((select_pass_kernel<1>), dim3(n_nodes), dim3(kThreads), 0, st, n_nodes, c.rowptr, c.col, c.heat, points, prefix, c.hist, c.top3);
    else if (level == 2) hipLaunchKernelGGL((select_pass_kernel<2>), dim3(n_nodes), dim3(kThreads), 0, st, n_nodes, c.rowptr, c.col, c.heat, points, prefix, c.hist, c.top3);
    else hipLaunchKernelGGL((select_pass_kernel<3>), dim3(n_nodes), dim3(kThreads), 0, st, n_nodes, c.rowptr, c.col, c.heat, points, prefix, c.hist, c.top3);
    er = hipMemcpyAsync(h.data(), c.hist, 8 * 2048, hipMemcpyDeviceToHost, st);
    if (er == hipSuccess) er = hipStreamSynchronize(st);
    if (er != hipSuccess) return fail_hip("histogram", er);
    const int bins = level == 3 ? 1024 : 2048;
    if (level == 1) {
      for (int b = 0; b < bins; ++b) positives += (long long)h[b];
      if (k_req > positives)
        return set_error(DIFUSCO_EINVAL, "mcts_heatmap: fewer positive entries (%lld) than expected_valid_value_num (%lld); "
                         "the reference raises IndexError here", positives, k_req);
      if (positives == 0) return set_error(DIFUSCO_EINVAL, "mcts_heatmap: IndexError: no positive entry (the reference indexes an empty valid_values array here)");
      // k == 0: the reference takes valid_values[-0] = valid_values[0], the SMALLEST positive value
      rank = k_req > 0 ? k_req : positives;
    }
    long long above = 0;
    int b = bins - 1;
    for (; b >= 0; --b) {
      if (above + (long long)h[b] >= rank) break;
      above += (long long)h[b];
    }
    if (b < 0) return set_error(DIFUSCO_EHIP, "mcts_heatmap: radix select lost its rank (level %d)", level);
    rank -= above;
    prefix = level == 3 ? ((prefix << 10) | (unsigned)b) : ((prefix << 11) | (unsigned)b);
  }
  float thr;
  std::memcpy(&thr, &prefix, 4);

  std::vector<int> leaves, ops;
  build_sum_program(n_nodes, leaves, ops);
  if (leaves.size() > 2 * ((size_t)n_nodes / 8 + 64) || ops.size() > 3 * ((size_t)n_nodes / 8 + 64))
    return set_error(DIFUSCO_EHIP, "mcts_heatmap: summation program larger than its buffer");
  er = hipMemcpyAsync(c.leaves, leaves.data(), 4 * leaves.size(), hipMemcpyHostToDevice, st);
  if (er == hipSuccess) er = hipMemcpyAsync(c.ops, ops.data(), 4 * ops.size(), hipMemcpyHostToDevice, st);
  Header hd{};
  hd.n = n_nodes; hd.prepared = 1; hd.n_edges = n_edges; hd.threshold = thr;
  hd.n_leaves = (int)leaves.size() / 2; hd.n_ops = (int)ops.size() / 3; hd.k_requested = k_req; hd.positives = positives;
  if (er == hipSuccess) er = hipMemcpyAsync(c.hdr, &hd, sizeof(hd), hipMemcpyHostToDevice, st);
  if (er == hipSuccess) er = hipStreamSynchronize(st);
  if (er != hipSuccess) return fail_hip("upload", er);
  if (threshold_out) *threshold_out = thr;
  return DIFUSCO_OK;
}

int difusco_host_rowsum_f32(const float* a, int n, float* out) {
  using namespace difusco;
  if (!a || !out || n < 1) return set_error(DIFUSCO_EINVAL, "host_rowsum_f32: bad arguments");
  std::vector<int> leaves, ops;
  build_sum_program(n, leaves, ops);
  std::vector<float> slot(leaves.size() / 2 + ops.size() / 3);
  for (size_t t = 0; t < leaves.size() / 2; ++t) {
    const float* x = a + leaves[2 * t];
    const int len = leaves[2 * t + 1];
    volatile float res;                      // (volatile: every partial sum is rounded to float32, whatever the host flags)
    if (len < 8) {
      res = 0.0f;
      for (int k = 0; k < len; ++k) res = res + x[k];
    } else {
      volatile float r[8];
      for (int k = 0; k < 8; ++k) r[k] = x[k];
      int p = 8;
      for (; p < len - (len % 8); p += 8)
        for (int k = 0; k < 8; ++k) r[k] = r[k] + x[p + k];
      volatile float s01 = r[0] + r[1], s23 = r[2] + r[3], s45 = r[4] + r[5], s67 = r[6] + r[7];
      volatile float s0123 = s01 + s23, s4567 = s45 + s67;
      res = s0123 + s4567;
      for (; p < len; ++p) res = res + x[p];
    }
    slot[t] = res;
  }
  for (size_t o = 0; o < ops.size() / 3; ++o) {
    volatile float v = (ops[3 * o + 1] < 0 ? 0.0f : slot[ops[3 * o + 1]]) + slot[ops[3 * o + 2]];
    slot[ops[3 * o]] = v;
  }
  *out = slot[ops[ops.size() - 3]];
  return DIFUSCO_OK;
}

int difusco_mcts_heatmap_rows(int n_nodes, int64_t n_edges, const float* points, const void* workspace, size_t workspace_bytes,
                              int row_begin, int row_count, float* out_rows, void* stream) {
  using namespace difusco;
  if (n_nodes < 1 || !points || !workspace || !out_rows || row_begin < 0 || row_count < 0 || row_begin + row_count > n_nodes)
    return set_error(DIFUSCO_EINVAL, "mcts_heatmap_rows: bad arguments");
  if (row_count == 0) return DIFUSCO_OK;
  Carve c;
  hipError_t er = carve(const_cast<void*>(workspace), n_nodes, n_edges, &c);
  if (er != hipSuccess) return fail_hip("rocprim size query", er);
  if (c.total > workspace_bytes) return set_error(DIFUSCO_EWORKSPACE, "mcts_heatmap: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Header hd;
  er = hipMemcpyAsync(&hd, c.hdr, sizeof(hd), hipMemcpyDeviceToHost, st);
  if (er == hipSuccess) er = hipStreamSynchronize(st);
  if (er != hipSuccess) return fail_hip("header", er);
  if (hd.prepared != 1 || hd.n != n_nodes || hd.n_edges != n_edges)
    return set_error(DIFUSCO_EINVAL, "mcts_heatmap_rows: the workspace was not prepared for this instance");
  const size_t lds = sizeof(float) * ((size_t)n_nodes + hd.n_leaves + hd.n_ops);
  static std::atomic<unsigned long long> attr_devices{0};
  er = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&rows_kernel), 160 * 1024);
  if (er != hipSuccess) return fail_hip("hipFuncSetAttribute", er);
  if (lds > 160 * 1024) return set_error(DIFUSCO_EUNSUPPORTED, "mcts_heatmap: row does not fit LDS");
  hipLaunchKernelGGL(rows_kernel, dim3(row_count), dim3(kThreads), lds, st, n_nodes, row_begin, c.rowptr, c.col, c.heat, c.rowptr_t,
                     c.col_t, c.heat_t, points, c.top3, hd.threshold, c.leaves, hd.n_leaves, c.ops, hd.n_ops, out_rows);
  er = hipGetLastError();
  if (er != hipSuccess) return fail_hip("rows_kernel", er);
  return DIFUSCO_OK;
}

}  // extern "C"
