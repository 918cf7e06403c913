// k-NN graph construction in the reference's layout:
// difusco/co_datasets/tsp_graph_dataset.py:53-62 - sklearn KDTree(points).query(points, k) on float64 coordinates,
// edge_index[0] = i repeated k times, edge_index[1] = the k nearest neighbours of i in increasing distance, self first.
//
// Brute force, one workgroup per query point: the N squared distances (float64, (dx*dx) + (dy*dy) without fused
// multiply-add, the arithmetic of the KD-tree's reduced distance) go to LDS (global scratch for N > 16000), an
// 8-bit-digit radix select finds the k-th smallest key, the k selected (distance, index) pairs are compacted and
// bitonic-sorted in LDS (ties: lower index first), and written straight into the [2, n k] int64 edge_index slice.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/difusco_hip.h"
#include "kernels.h"

namespace difusco {
namespace {

constexpr int KNN_THREADS = 256;
constexpr int KNN_MAX_K = 1024;
constexpr int KNN_LDS_POINTS = 16000;     // 16000 * 8 B = 125 KiB of keys in LDS (+ 16 KiB sort buffer + histogram)

__device__ __forceinline__ unsigned long long dist_key(double xi, double yi, double xj, double yj) {
  const double dx = xi - xj, dy = yi - yj;
  const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
  return (unsigned long long)__double_as_longlong(d);      // d >= 0: the bit pattern orders like the value
}

// keys: LDS or global scratch of this block (n entries)
template <bool IN_LDS>
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const double* __restrict__ points, int n, int k, long long node_offset,
                                                          long long* __restrict__ row0, long long* __restrict__ row1,
                                                          unsigned long long* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: [sort keys KP][sort idx KP][hist 256][misc 8] then (IN_LDS) the n distance keys
  int kp = 2;
  while (kp < k) kp <<= 1;
  unsigned long long* skey = reinterpret_cast<unsigned long long*>(smem);
  int* sidx = reinterpret_cast<int*>(skey + kp);
  unsigned* hist = reinterpret_cast<unsigned*>(sidx + kp);
  unsigned* misc = hist + 256;        // [0] selected count, [2] number of keys equal to the k-th
  unsigned long long* keys = IN_LDS ? reinterpret_cast<unsigned long long*>(misc + 8) : scratch + (long long)blockIdx.x * n;
  const int tid = threadIdx.x;

  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const double xi = points[2 * i], yi = points[2 * i + 1];
    for (int j = tid; j < n; j += KNN_THREADS) keys[j] = dist_key(xi, yi, points[2 * j], points[2 * j + 1]);
    __syncthreads();
    // ---- radix select: prefix = the digits fixed so far of the k-th smallest key ----
    unsigned long long prefix = 0, mask = 0;
    int want = k;                      // rank (1-based) of the wanted key among the keys matching the prefix
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int j = tid; j < n; j += KNN_THREADS) {
        const unsigned long long v = keys[j];
        if ((v & mask) == prefix) atomicAdd(&hist[(unsigned)(v >> shift) & 255u], 1u);
      }
      __syncthreads();
      // every thread walks the 256-bin histogram identically (cheap, avoids another barrier round)
      int acc = 0, digit = 0;
      for (int d = 0; d < 256; ++d) {
        const int c = (int)hist[d];
        if (acc + c >= want) {
          digit = d;
          break;
        }
        acc += c;
      }
      want -= acc;
      prefix |= (unsigned long long)digit << shift;
      mask |= 255ULL << shift;
      __syncthreads();
    }
    const unsigned long long kth = prefix;   // the k-th smallest key; `want` of the keys equal to it are taken
    // ---- compact: everything below kth, then the lowest-index ties ----
    if (tid == 0) {
      misc[0] = 0;
      misc[1] = 0;
    }
    for (int s = tid; s < kp; s += KNN_THREADS) {
      skey[s] = ~0ULL;
      sidx[s] = 0x7fffffff;
    }
    __syncthreads();
    for (int j = tid; j < n; j += KNN_THREADS) {
      const unsigned long long v = keys[j];
      if (v < kth) {
        const unsigned s = atomicAdd(&misc[0], 1u);
        skey[s] = v;
        sidx[s] = j;
      }
    }
    __syncthreads();
    // keys equal to kth: the `want` lowest indices are taken (want >= 1: the k-th key itself).  Every thread collects the
    // ties of its strided share into a list (the histogram's LDS, free now); there are almost always exactly `want` = 1
    // of them.  One thread orders the list by index and appends the first `want`; only when more than 256 keys tie does
    // it fall back to the serial scan over all n keys.
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += KNN_THREADS)
      if (keys[j] == kth) {
        const unsigned t = atomicAdd(&misc[2], 1u);
        if (t < 256u) hist[t] = (unsigned)j;
      }
    __syncthreads();
    if (tid == 0) {
      unsigned s = misc[0];
      const unsigned nt = misc[2];
      if (nt <= 256u) {
        for (unsigned a = 1; a < nt; ++a) {                 // insertion sort of the (tiny) tie list by index
          const unsigned v = hist[a];
          unsigned b = a;
          for (; b > 0 && hist[b - 1] > v; --b) hist[b] = hist[b - 1];
          hist[b] = v;
        }
        for (int t = 0; t < want && t < (int)nt; ++t) {
          skey[s] = kth;
          sidx[s] = (int)hist[t];
          ++s;
        }
      } else {
        int taken = 0;
        for (int j = 0; j < n && taken < want; ++j)
          if (keys[j] == kth) {
            skey[s] = kth;
            sidx[s] = j;
            ++s;
            ++taken;
          }
      }
    }
    __syncthreads();
    // ---- bitonic sort of kp (key, index) pairs ----
    for (int size = 2; size <= kp; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < kp / 2; t += KNN_THREADS) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const bool up = (lo & size) == 0;
          const unsigned long long a = skey[lo], b = skey[hi];
          const int ia = sidx[lo], ib = sidx[hi];
          const bool gt = a > b || (a == b && ia > ib);
          if (gt == up) {
            skey[lo] = b;
            skey[hi] = a;
            sidx[lo] = ib;
            sidx[hi] = ia;
          }
        }
        __syncthreads();
      }
    for (int r = tid; r < k; r += KNN_THREADS) {
      row0[(long long)i * k + r] = node_offset + i;
      row1[(long long)i * k + r] = node_offset + sidx[r];
    }
    __syncthreads();
  }
}

size_t knn_fixed_lds(int k) {
  int kp = 2;
  while (kp < k) kp <<= 1;
  return (size_t)kp * 12 + 256 * 4 + 32;
}

}  // namespace
}  // namespace difusco

extern "C" {

int difusco_knn_graph_workspace_bytes(int n_nodes, int k, size_t* bytes) {
  using namespace difusco;
  if (!bytes || n_nodes < 1 || k < 1 || k > n_nodes || k > KNN_MAX_K)
    return set_error(DIFUSCO_EINVAL, "knn_graph: needs 1 <= k <= min(n_nodes, %d)", KNN_MAX_K);
  *bytes = n_nodes <= KNN_LDS_POINTS ? 256 : (size_t)1024 * n_nodes * 8;
  return DIFUSCO_OK;
}

int difusco_knn_graph(int n_nodes, int k, const double* points, int64_t node_offset, int64_t* edge_row0,
                      int64_t* edge_row1, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace difusco;
  size_t need = 0;
  const int rc = difusco_knn_graph_workspace_bytes(n_nodes, k, &need);
  if (rc != DIFUSCO_OK) return rc;
  if (!points || !edge_row0 || !edge_row1) return set_error(DIFUSCO_EINVAL, "knn_graph: null device array");
  hipStream_t st = (hipStream_t)stream;
  const size_t fixed = (knn_fixed_lds(k) + 15) / 16 * 16;
  if (n_nodes <= KNN_LDS_POINTS) {
    const size_t lds = fixed + (size_t)n_nodes * 8;
    static std::atomic<unsigned long long> attr_devices{0};      // per device (a process-wide flag skipped device 2+)
    {
      hipError_t er = difusco::ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&knn_kernel<true>), 160 * 1024);
      if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "knn_graph: %s", hipGetErrorString(er));
    }
    hipLaunchKernelGGL(knn_kernel<true>, dim3(n_nodes), dim3(KNN_THREADS), lds, st, points, n_nodes, k,
                       (long long)node_offset, (long long*)edge_row0, (long long*)edge_row1, (unsigned long long*)nullptr);
  } else {
    if (!workspace || workspace_bytes < need) return set_error(DIFUSCO_EINVAL, "knn_graph: workspace %zu < %zu bytes", workspace_bytes, need);
    hipLaunchKernelGGL(knn_kernel<false>, dim3(1024), dim3(KNN_THREADS), fixed, st, points, n_nodes, k,
                       (long long)node_offset, (long long*)edge_row0, (long long*)edge_row1, (unsigned long long*)workspace);
  }
  hipError_t er = hipGetLastError();
  if (er != hipSuccess) return set_error(DIFUSCO_EHIP, "knn_graph launch: %s", hipGetErrorString(er));
  return DIFUSCO_OK;
}

}  // extern "C"
