// Standalone check of the LDS-DMA stage image against the register-staged image (same swizzle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int wslot(int entry, int half) { return entry * 16 + ((half ^ ((entry >> 3) & 1)) << 3); }
constexpr int PLANE = 256 * 16, BUF = 2 * PLANE;
__global__ void k(const unsigned short* __restrict__ planes, long long plane_stride, int shape, unsigned* mismatches,
                  unsigned short* dump) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wbuf = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // register path -> buffer 1
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i, entry = c >> 1, half = c & 1;
    const unsigned vo = shape == 1 ? (entry >> 8) * 4096 + (entry & 255) * 16 + half * 8
                                   : (entry >> 6) * 4096 + (entry & 63) * 16 + half * 8;
    const v4u a = *reinterpret_cast<const v4u*>(planes + vo);
    const v4u b = *reinterpret_cast<const v4u*>(planes + plane_stride + vo);
    *reinterpret_cast<v4u*>(wbuf + BUF + wslot(entry, half)) = a;
    *reinterpret_cast<v4u*>(wbuf + BUF + PLANE + wslot(entry, half)) = b;
  }
  // DMA path -> buffer 0
  const int entry0 = (2 * wave) * 32 + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
  const unsigned dvoff1 = (entry0 >> 8) * 4096 + (entry0 & 255) * 16 + half * 8;
  const unsigned dvoff2 = (entry0 >> 6) * 4096 + (entry0 & 63) * 16 + half * 8;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* src = planes + pl * plane_stride + (shape == 1 ? dvoff1 : dvoff2) + i * 512;
      unsigned short* dst = wbuf + pl * PLANE + (2 * wave + i) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  unsigned bad = 0;
  for (int x = tid; x < BUF; x += 256) {
    if (wbuf[x] != wbuf[BUF + x]) ++bad;
    dump[x] = wbuf[x];
    dump[BUF + x] = wbuf[BUF + x];
  }
  atomicAdd(mismatches, bad);
}
int main() {
  const long long stride = 256 * 256;
  std::vector<unsigned short> h(2 * stride);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(i * 2654435761u >> 16);
  unsigned short *d, *dump;
  unsigned* mm;
  hipMalloc(&d, h.size() * 2);
  hipMalloc(&dump, 2 * BUF * 2);
  hipMalloc(&mm, 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int shape = 1; shape <= 2; ++shape) {
    hipMemset(mm, 0, 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 2 * BUF * 2, 0, d, stride, shape, mm, dump);
    unsigned r = 0;
    hipMemcpy(&r, mm, 4, hipMemcpyDeviceToHost);
    std::vector<unsigned short> img(2 * BUF);
    hipMemcpy(img.data(), dump, img.size() * 2, hipMemcpyDeviceToHost);
    int first = -1;
    for (int x = 0; x < BUF; ++x) if (img[x] != img[BUF + x]) { first = x; break; }
    printf("shape %d: %u mismatching elements of %d (first at %d)  err=%s\n", shape, r, BUF, first, hipGetErrorString(hipGetLastError()));
    if (first >= 0) {
      for (int x = first; x < first + 16; ++x) printf(" %04x/%04x", img[x], img[BUF + x]);
      printf("\n");
    }
  }
  return 0;
}
