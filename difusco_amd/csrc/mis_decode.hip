// Greedy MIS decode from node scores (difusco/utils/mis_utils.py:3-18 `mis_decode_np`, called per sample at
// pl_mis_model.py:196), SURVEY 8(f)-3.
//
// The reference visits the nodes in decreasing score and takes a node unless an earlier-taken neighbour excluded it:
// the lexicographically first maximal independent set of that order.  That set has a parallel characterisation -
// a node is IN iff all its higher-priority neighbours are OUT, and OUT iff one of them is IN - so it is computed
// in rounds over all nodes at once (a node decides as soon as its higher-priority neighbours have); the number of
// rounds is the longest priority-decreasing dependency chain, a few dozen on Erdos-Renyi graphs.
//   1. rank: stable radix sort (rocPRIM) of (-score, index) - ties keep index order (numpy's argsort is unstable
//      there; fixtures are tie-free);
//   2. rounds of mis_round_kernel until no node is undecided (device-side counter, polled every few rounds).
// Works on the CSR of the (symmetric, self loops allowed) adjacency of the whole call; graphs of a batch are
// independent components, so one call decodes all of them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/difusco_hip.h"
#include "kernels.h"

namespace difusco {
namespace {

__global__ void mis_iota_kernel(int n, unsigned* __restrict__ idx) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) idx[v] = (unsigned)v;
}

__global__ void mis_rank_kernel(int n, const unsigned* __restrict__ order, int* __restrict__ rank, int* __restrict__ state) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  rank[order[p]] = p;
  state[order[p]] = 0;
}

// state: 0 undecided, 1 in, 2 out.  One wavefront per node: lanes stride over its neighbour list.
__global__ __launch_bounds__(256) void mis_round_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                        const int* __restrict__ rank, int* __restrict__ state,
                                                        unsigned* __restrict__ undecided) {
  const int v = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (v >= n) return;
  if (state[v] != 0) return;                 // wave uniform
  const int rv = rank[v];
  int any_in = 0, any_open = 0;
  for (int e = rowptr[v] + lane; e < rowptr[v + 1]; e += 64) {
    const int u = col[e];
    if (u == v) continue;
    if (rank[u] < rv) {
      const int su = state[u];               // a stale read only delays the decision by a round
      any_in |= su == 1;
      any_open |= su == 0;
    }
  }
  any_in = __any(any_in);
  any_open = __any(any_open);
  if (lane == 0) {
    if (any_in) state[v] = 2;
    else if (!any_open) state[v] = 1;
    else atomicAdd(undecided, 1u);
  }
}

__global__ void mis_finish_kernel(int n, const int* __restrict__ state, int* __restrict__ solution) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) solution[v] = state[v] == 1 ? 1 : 0;
}

size_t up256(size_t x) { return (x + 255) / 256 * 256; }

struct MisCarve {
  float *key_a, *key_b;
  unsigned *idx_a, *idx_b, *counter;
  int *rank, *state;
  void* temp;
  size_t temp_bytes, total;
};

hipError_t mis_carve(void* base, int n, MisCarve* c) {
  size_t t = 0;
  hipError_t er = rocprim::radix_sort_pairs_desc(nullptr, t, (float*)nullptr, (float*)nullptr, (unsigned*)nullptr,
                                                 (unsigned*)nullptr, (size_t)n, 0, 32, 0, false);
  if (er != hipSuccess) return er;
  c->temp_bytes = t;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    size_t at = cur;
    cur += up256(bytes);
    return base ? (void*)((char*)base + at) : (void*)nullptr;
  };
  c->key_a = (float*)take(4 * (size_t)n);
  c->key_b = (float*)take(4 * (size_t)n);
  c->idx_a = (unsigned*)take(4 * (size_t)n);
  c->idx_b = (unsigned*)take(4 * (size_t)n);
  c->rank = (int*)take(4 * (size_t)n);
  c->state = (int*)take(4 * (size_t)n);
  c->counter = (unsigned*)take(256);
  c->temp = take(t);
  c->total = cur;
  return hipSuccess;
}

}  // namespace
}  // namespace difusco

extern "C" {

int difusco_mis_decode_workspace_bytes(int n_nodes, size_t* bytes) {
  using namespace difusco;
  if (!bytes || n_nodes < 1) return set_error(DIFUSCO_EINVAL, "mis_decode_workspace_bytes: bad arguments");
  MisCarve c;
  hipError_t er = mis_carve(nullptr, n_nodes, &c);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "rocprim temp size: %s", hipGetErrorString(er));
  *bytes = c.total;
  return DIFUSCO_OK;
}

int difusco_mis_decode(int n_nodes, const int32_t* rowptr, const int32_t* col, const float* scores, int32_t* solution,
                       void* workspace, size_t workspace_bytes, int32_t* rounds_out, void* stream) {
  using namespace difusco;
  if (n_nodes < 1 || !rowptr || !col || !scores || !solution || !workspace)
    return set_error(DIFUSCO_EINVAL, "mis_decode: null device array or empty graph");
  MisCarve c;
  hipError_t er = mis_carve(workspace, n_nodes, &c);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "rocprim temp size: %s", hipGetErrorString(er));
  if (workspace_bytes < c.total) return set_error(DIFUSCO_EINVAL, "mis_decode: workspace %zu < %zu bytes", workspace_bytes, c.total);
  hipStream_t st = (hipStream_t)stream;
  const int n = n_nodes;
  const unsigned g1 = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(mis_iota_kernel, dim3(g1), dim3(256), 0, st, n, c.idx_a);
  size_t tb = c.temp_bytes;
  // descending by score; the sort is stable, so equal scores keep increasing index order
  er = rocprim::radix_sort_pairs_desc(c.temp, tb, scores, c.key_b, c.idx_a, c.idx_b, (size_t)n, 0, 32, st, false);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "radix_sort_pairs_desc: %s", hipGetErrorString(er));
  hipLaunchKernelGGL(mis_rank_kernel, dim3(g1), dim3(256), 0, st, n, c.idx_b, c.rank, c.state);
  const unsigned gw = (unsigned)(((long long)n * 64 + 255) / 256);
  int rounds = 0;
  unsigned left = 1;
  while (left != 0) {
    for (int r = 0; r < 4; ++r) {            // the counter of the last round of the group decides
      er = hipMemsetAsync(c.counter, 0, 4, st);
      if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "memset: %s", hipGetErrorString(er));
      hipLaunchKernelGGL(mis_round_kernel, dim3(gw), dim3(256), 0, st, n, rowptr, col, c.rank, c.state, c.counter);
      ++rounds;
    }
    er = hipMemcpyAsync(&left, c.counter, 4, hipMemcpyDeviceToHost, st);
    if (er == hipSuccess) er = hipStreamSynchronize(st);
    if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "mis_decode rounds: %s", hipGetErrorString(er));
    if (rounds > 4 * n_nodes + 8) return set_error(DIFUSCO_EINVAL, "mis_decode: no progress (adjacency not symmetric?)");
  }
  hipLaunchKernelGGL(mis_finish_kernel, dim3(g1), dim3(256), 0, st, n, c.state, solution);
  er = hipStreamSynchronize(st);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "mis_decode: %s", hipGetErrorString(er));
  if (rounds_out) *rounds_out = rounds;
  return DIFUSCO_OK;
}

}  // extern "C"
