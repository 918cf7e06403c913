// buffer_store through a per-wave 32 KiB resource: which descriptor / cache-policy combination writes what?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int AUX, int BIG, int GUARD>
__global__ void k(float* e, int n_valid) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x * 4 + wave;
  float* etile = e + (long long)tile * 8192;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(etile, 0, BIG ? 0x7fffffff : 32768, 0x00020000);
  const bool valid = !GUARD || (tile * 32 + (lane & 31)) < n_valid;
  if (valid) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, s * 1024, AUX);
      r[0] += 1; r[1] += 2; r[2] += 3; r[3] += 4;
      __builtin_amdgcn_raw_buffer_store_b128(r, rs, lane * 16, s * 1024, AUX);
    }
  }
}
template <int AUX, int BIG, int GUARD>
int run(const char* name) {
  const int tiles = 64 * 4; const size_t n = (size_t)tiles * 8192;
  std::vector<unsigned> h(n); for (size_t i = 0; i < n; ++i) h[i] = (unsigned)i;
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  const int n_valid = tiles * 32 - 37;
  hipLaunchKernelGGL((k<AUX, BIG, GUARD>), dim3(64), dim3(256), 0, 0, d, n_valid);
  hipDeviceSynchronize();
  std::vector<unsigned> o(n); hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const size_t tile = i / 8192, in = i % 8192, lane = (in % 256) / 4, row = tile * 32 + (lane & 31);
    const bool valid = !GUARD || row < (size_t)n_valid;
    const unsigned want = (unsigned)i + (valid ? (unsigned)(i % 4) + 1 : 0);
    if (o[i] != want) ++bad;
  }
  printf("%s: %zu wrong of %zu\n", name, bad, n);
  hipFree(d);
  return bad != 0;
}
int main() {
  int r = 0;
  r |= run<0, 1, 0>("aux0 big");
  r |= run<0, 0, 0>("aux0 32K");
  r |= run<2, 1, 0>("nt big");
  r |= run<2, 0, 0>("nt 32K");
  r |= run<2, 0, 1>("nt 32K guarded");
  r |= run<0, 0, 1>("aux0 32K guarded");
  return r;
}
