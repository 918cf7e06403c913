// Batched 2-opt refinement of decoded tours (difusco/utils/tsp_utils.py:12-49 `batched_two_opt_torch`), SURVEY 8(f)-2.
//
// The reference materialises four N x N float64 distance matrices per iteration (A_ij, A_i+1,j+1, A_i,i+1, A_j,j+1),
// takes triu(diagonal=2) of their combination, and an argmin per tour; up to 1000-5000 iterations.  Here one
// iteration is three small launches with nothing N x N in memory:
//   two_opt_prep_kernel   P[k] = points[tour[k]] (tour order) and d[k] = |P[k] - P[k+1]|;
//   two_opt_best_kernel   every thread keeps P[j], P[j+1], d[j] of its columns in registers, a block sweeps a tile of
//                         rows i, change = ((A_ij + A_i+1,j+1) - d_i) - d_j in float64 in the reference's operation
//                         order, running (min, first flat index) per thread -> block -> partial[];
//   two_opt_apply_kernel  per tour: argmin over the partials (ties: lowest flat index = first occurrence, what
//                         torch.argmin returns on the flattened matrix); min over the batch decides (tsp_utils.py:
//                         33,39: one global `min_change < -1e-6` test, every tour applies its own best move);
//                         the segment tour[i+1 .. j] is reversed in place, the iteration counter advances.
// Invalid entries (j < i + 2) are zeros in the reference's matrix, so the minimum is never positive and the argmin of
// an all-non-improving matrix is flat index 0 = a no-op reversal: the running best starts at (0.0, 0).
// The host only polls a `done` flag every few iterations; once set, the remaining launches return immediately.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/difusco_hip.h"
#include "kernels.h"

namespace difusco {
namespace {

// float64 distance exactly as torch evaluates sqrt(sum((p - q) ** 2, -1)): two products, one sum, no fused multiply-add
__device__ __forceinline__ double dist2d(double dx, double dy) { return sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))); }

struct Best {
  double v;
  long long idx;
};

__device__ __forceinline__ bool better(double v, long long idx, const Best& b) { return v < b.v || (v == b.v && idx < b.idx); }

struct TwoOptState {       // device-resident loop state
  int done;
  int pad;
  long long iterations;
};

__global__ void two_opt_prep_kernel(const double* __restrict__ points, const int* __restrict__ tours, int n, int batch,
                                    double2* __restrict__ tp, double* __restrict__ dlen, const TwoOptState* st) {
  if (st->done) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (k > n) return;
  const int* tour = tours + (long long)b * (n + 1);
  const int c = tour[k];
  const double2 p = make_double2(points[2 * c], points[2 * c + 1]);
  tp[(long long)b * (n + 1) + k] = p;
  if (k < n) {
    const int c1 = tour[k + 1];
    const double dx = p.x - points[2 * c1], dy = p.y - points[2 * c1 + 1];
    dlen[(long long)b * n + k] = dist2d(dx, dy);                // A_i,i+1 (tsp_utils.py:28)
  }
}

constexpr int TI = 16;     // rows per block
constexpr int JPT = 4;     // columns per thread per sweep (256 threads x 4 = 1024 columns per sweep)

__global__ __launch_bounds__(256) void two_opt_best_kernel(const double2* __restrict__ tp, const double* __restrict__ dlen,
                                                           int n, Best* __restrict__ partial, const TwoOptState* st) {
  if (st->done) return;
  const int b = blockIdx.y, i0 = blockIdx.x * TI;
  const double2* P = tp + (long long)b * (n + 1);
  const double* D = dlen + (long long)b * n;
  __shared__ double2 pi[TI + 1];
  __shared__ double di[TI];
  for (int t = threadIdx.x; t <= TI; t += blockDim.x)
    if (i0 + t <= n) pi[t] = P[i0 + t];
  for (int t = threadIdx.x; t < TI; t += blockDim.x)
    if (i0 + t < n) di[t] = D[i0 + t];
  __syncthreads();
  Best best{0.0, 0};
  const int rows = (n - i0) < TI ? (n - i0) : TI;
  // columns j >= i0 + 2 matter for this tile; sweep them in chunks of 256 * JPT
  for (int jbase = i0 + 2; jbase < n; jbase += 256 * JPT) {
    double2 pj[JPT], pj1[JPT];
    double dj[JPT];
    int jj[JPT];
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
      const int j = jbase + u * 256 + threadIdx.x;
      jj[u] = j;
      if (j < n) {
        pj[u] = P[j];
        pj1[u] = P[j + 1];
        dj[u] = D[j];
      }
    }
    for (int r = 0; r < rows; ++r) {
      const int i = i0 + r;
      const double2 a = pi[r], a1 = pi[r + 1];
      const double d_i = di[r];
#pragma unroll
      for (int u = 0; u < JPT; ++u) {
        const int j = jj[u];
        if (j < n && j >= i + 2) {
          const double x0 = a.x - pj[u].x, y0 = a.y - pj[u].y;
          const double x1 = a1.x - pj1[u].x, y1 = a1.y - pj1[u].y;
          // change = A_ij + A_i+1,j+1 - A_i,i+1 - A_j,j+1, evaluated left to right (tsp_utils.py:31)
          const double change = __dsub_rn(__dsub_rn(__dadd_rn(dist2d(x0, y0), dist2d(x1, y1)), d_i), dj[u]);
          const long long idx = (long long)i * n + j;
          if (better(change, idx, best)) best = Best{change, idx};
        }
      }
    }
  }
  // block reduction (min value, then lowest flat index)
  __shared__ Best red[256];
  red[threadIdx.x] = best;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s && better(red[threadIdx.x + s].v, red[threadIdx.x + s].idx, red[threadIdx.x])) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long long)b * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void two_opt_apply_kernel(int* __restrict__ tours, int n, int batch, int nblk,
                                                            const Best* __restrict__ partial, long long max_iterations,
                                                            TwoOptState* st, Best* __restrict__ chosen) {
  if (st->done) return;
  __shared__ Best red[256];
  __shared__ double gmin;
  // per tour: argmin over its partials
  for (int b = 0; b < batch; ++b) {
    Best best{0.0, 0};
    for (int t = threadIdx.x; t < nblk; t += blockDim.x) {
      const Best c = partial[(long long)b * nblk + t];
      if (better(c.v, c.idx, best)) best = c;
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s && better(red[threadIdx.x + s].v, red[threadIdx.x + s].idx, red[threadIdx.x])) red[threadIdx.x] = red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) chosen[b] = red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double m = chosen[0].v;
    for (int b = 1; b < batch; ++b) m = chosen[b].v < m ? chosen[b].v : m;
    gmin = m;
  }
  __syncthreads();
  if (!(gmin < -1e-6)) {                                  // tsp_utils.py:39,44
    if (threadIdx.x == 0) st->done = 1;
    return;
  }
  for (int b = 0; b < batch; ++b) {
    const long long idx = chosen[b].idx;
    const int mi = (int)(idx / n), mj = (int)(idx % n);
    int* tour = tours + (long long)b * (n + 1);
    const int len = mj - mi;                              // tour[mi+1 .. mj] reversed (tsp_utils.py:41)
    for (int t = threadIdx.x; t < len / 2; t += blockDim.x) {
      const int x = mi + 1 + t, y = mj - t;
      const int tmp = tour[x];
      tour[x] = tour[y];
      tour[y] = tmp;
    }
  }
  if (threadIdx.x == 0) {
    st->iterations += 1;
    if (st->iterations >= max_iterations) st->done = 1;   // tsp_utils.py:46-47
  }
}

size_t up256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace
}  // namespace difusco

extern "C" {

int difusco_tsp_two_opt_workspace_bytes(int n_nodes, int batch, size_t* bytes) {
  using namespace difusco;
  if (!bytes || n_nodes < 4 || batch < 1) return set_error(DIFUSCO_EINVAL, "two_opt_workspace_bytes: bad arguments");
  const size_t nblk = (size_t)(n_nodes + TI - 1) / TI;
  *bytes = up256(sizeof(double2) * (size_t)batch * (n_nodes + 1)) + up256(sizeof(double) * (size_t)batch * n_nodes) +
           up256(sizeof(Best) * (size_t)batch * nblk) + up256(sizeof(Best) * (size_t)batch) + 256;
  return DIFUSCO_OK;
}

int difusco_tsp_two_opt(int n_nodes, int batch, const double* points, int32_t* tours, int64_t max_iterations,
                        void* workspace, size_t workspace_bytes, int64_t* iterations_out, void* stream) {
  using namespace difusco;
  if (n_nodes < 4 || batch < 1 || !points || !tours || !workspace || max_iterations < 0)
    return set_error(DIFUSCO_EINVAL, "tsp_two_opt: needs n_nodes >= 4, batch >= 1 and non-null device arrays");
  size_t need = 0;
  difusco_tsp_two_opt_workspace_bytes(n_nodes, batch, &need);
  if (workspace_bytes < need) return set_error(DIFUSCO_EINVAL, "tsp_two_opt: workspace %zu < %zu bytes", workspace_bytes, need);
  const int n = n_nodes, nblk = (n + TI - 1) / TI;
  char* w = (char*)workspace;
  double2* tp = (double2*)w;
  w += up256(sizeof(double2) * (size_t)batch * (n + 1));
  double* dlen = (double*)w;
  w += up256(sizeof(double) * (size_t)batch * n);
  Best* partial = (Best*)w;
  w += up256(sizeof(Best) * (size_t)batch * nblk);
  Best* chosen = (Best*)w;
  w += up256(sizeof(Best) * (size_t)batch);
  TwoOptState* st = (TwoOptState*)w;
  hipStream_t s = (hipStream_t)stream;
  hipError_t er = hipMemsetAsync(st, 0, sizeof(TwoOptState), s);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "memset: %s", hipGetErrorString(er));
  TwoOptState host{0, 0, 0};
  // the reference evaluates the moves once more after the last applied one and stops on `min_change >= -1e-6`;
  // at most max_iterations moves are applied (tsp_utils.py:20-47)
  const int poll = 8;
  const long long loops = max_iterations > 0 ? max_iterations : 1;   // the reference always evaluates (and may apply) once
  for (long long it = 0; it < loops; ++it) {
    hipLaunchKernelGGL(two_opt_prep_kernel, dim3((n + 1 + 255) / 256, batch), dim3(256), 0, s, points, tours, n, batch, tp,
                       dlen, st);
    hipLaunchKernelGGL(two_opt_best_kernel, dim3(nblk, batch), dim3(256), 0, s, tp, dlen, n, partial, st);
    hipLaunchKernelGGL(two_opt_apply_kernel, dim3(1), dim3(256), 0, s, tours, n, batch, nblk, partial,
                       (long long)max_iterations, st, chosen);
    if ((it + 1) % poll == 0 || it + 1 == loops) {
      er = hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s);
      if (er == hipSuccess) er = hipStreamSynchronize(s);
      if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "two_opt state: %s", hipGetErrorString(er));
      if (host.done) break;
    }
  }
  er = hipMemcpyAsync(&host, st, sizeof(host), hipMemcpyDeviceToHost, s);
  if (er == hipSuccess) er = hipStreamSynchronize(s);
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "two_opt state: %s", hipGetErrorString(er));
  if (iterations_out) *iterations_out = host.iterations;
  return DIFUSCO_OK;
}

}  // extern "C"
