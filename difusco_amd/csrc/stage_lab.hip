// Stage-loop laboratory (PROFILING library only): GEMM 1 of the fused edge layer - Ce = C e on the tiled e stream, fp16x3
// split products, weight planes streamed through LDS by LDS-DMA - and NOTHING else, in several workgroup geometries and
// synchronisation schemes.  It exists to answer one question with measurements before the production kernel
// (edge_layer_kernel.h) is touched: what does the weight-stage loop cost in each geometry ("GEMM-only
// ablation first each time")?  Same operand layouts as the production kernel (plane slabs of weights.py, wslot swizzle,
// tiled e), real results (scripts/bench_stage_lab.py checks them against a float64 product), so every variant is a
// drop-in candidate for the production stage loop.
//
// Template parameters of a variant:
//   EPW   edges per wave: 32 (one B tile per weight fragment, production) or 64 (two B tiles against every ds_read_b128
//         of a weight fragment: half the LDS reads and half the L2 -> LDS weight stream per edge, 48 MFMAs per stage)
//   WAVES waves per workgroup: 4 or 8 (8 waves share one weight stream: 256 edges per pass at 32 edges per wave)
//   NBUF  weight stage buffers (16 KiB each): 2 = double buffer (production), 3 / 4 = one / two stages in flight across
//         the stage barrier
//   SYNC  0: s_waitcnt vmcnt + __syncthreads() per stage (production);  1: raw s_barrier + counted vmcnt (the newest
//         requests stay in flight across the barrier);  2: no barrier - per-wave LDS flags ("stage s landed" / "stage s
//         read"), a wave starts stage t+1 when ITS data has landed, whatever its siblings are doing
//   RING  e slabs (register ring) in flight per tile
//   PRIO  1: s_setprio 1 around every MFMA group
//   MINB  workgroups per CU the register budget is sized for (__launch_bounds__ second argument): 2 -> <= 256 registers,
//         1 -> up to 512 (one wave per SIMD with 4-wave workgroups)
//   VMIX  synthetic EPILOGUE of the size of the fused kernel's (per 32-edge tile and lane 128 elements x 2 rounds of {fma, exp2,
//         add, rcp, mul, fma, max, add}: 2,048 vector + 512 transcendental instructions, 256 LDS operations on a wave-private
//         scratch), to measure how such work overlaps a GEMM.  1: SERIAL - after the 16 stages, on the tile's own accumulators (the
//         production structure: a wave alternates matrix and vector phases, two waves per SIMD overlap by chance).  2:
//         INTERLEAVED - one 64th of the epilogue of ANOTHER tile (an independent register array) behind every MFMA group of the
//         GEMM, one wave per SIMD (the hand-scheduled structure of DESIGN section 7 (c))
#include "edge_layer_common.h"

#ifdef DIFUSCO_PROFILING
#include "../../include/difusco_hip.h"

// The file is compiled twice into the profiling library: as is, and with -DDIFUSCO_LAB_NOPK under the target feature
// -packed-fp32-ops (the operand split then uses plain v_sub_f32 instead of v_pk_add_f32): difusco_lab_gemm1_nopk.
#ifdef DIFUSCO_LAB_NOPK
#define LAB_NS lab_nopk
#define LAB_ENTRY difusco_lab_gemm1_nopk
#else
#define LAB_NS lab
#define LAB_ENTRY difusco_lab_gemm1
#endif

namespace difusco {
namespace LAB_NS {

constexpr int H = 256;
constexpr int NS = 16;                 // stages = k slabs of C
constexpr int PLANE = 256 * 16;        // 16-bit elements per plane per stage
constexpr int BUF = 2 * PLANE;         // elements per stage buffer (two planes, 16 KiB)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter on gfx9");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// VMEM operations of one wave that may still be in flight at the end of stage t when everything up to and including the
// requests of weight stage t+1 must have landed (requests complete in order).  Issue order: e ring fill, weight stages
// 0 .. DIST-1 (prologue); then per stage s: weight stage s+DIST, e slab s+RING.
template <int TPW, int PPW, int DIST, int RING>
constexpr int allowed_in_flight(int t) {
  int pos = RING * 2 * TPW;
  int dma_end[NS + 8] = {};
  for (int s = 0; s < DIST && s < NS; ++s) {
    pos += PPW;
    dma_end[s] = pos;
  }
  for (int st = 0; st <= t; ++st) {
    if (st + DIST < NS) {
      pos += PPW;
      dma_end[st + DIST] = pos;
    }
    if (st + RING < NS) pos += 2 * TPW;
  }
  return pos - dma_end[t + 1];
}

// SYNC 2: spin until every one of the WAVES monotone counters at `f` (LDS, 16-byte aligned) has reached `need`
template <int WAVES>
__device__ __forceinline__ void lab_flags_wait(volatile int* f, int need) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  while (true) {
    v4i a = *reinterpret_cast<volatile v4i*>(f);
    int mn = min(min(a[0], a[1]), min(a[2], a[3]));
    if constexpr (WAVES == 8) {
      v4i b = *reinterpret_cast<volatile v4i*>(f + 4);
      mn = min(mn, min(min(b[0], b[1]), min(b[2], b[3])));
    }
    if (__builtin_amdgcn_readfirstlane(mn) >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int EPW, int WAVES, int NBUF, int SYNC, int RING, int PRIO, int MINB, int DIST, int VMIX>
__global__ __launch_bounds__(64 * WAVES, MINB) void stage_lab_kernel(const float* __restrict__ e,
                                                                     const unsigned short* __restrict__ c_planes,
                                                                     long long plane_stride, float* __restrict__ out,
                                                                     int n_tiles, float inv_c, int do_store) {
  typedef FFp16 T;
  typedef typename T::frag frag;
  constexpr int TPW = EPW / 32;          // 32-edge tiles per wave
  constexpr int PP = 8 / WAVES;          // 1 KiB LDS-DMA pieces per wave, plane and stage
  constexpr int PPW = 2 * PP;            // ... per wave and stage
  static_assert(EPW == 32 || EPW == 64, "");
  static_assert(WAVES == 4 || WAVES == 8, "");
  static_assert(SYNC != 0 || NBUF == 2, "the __syncthreads() scheme is the double buffer");
  // DIST = stages a weight request is issued ahead of its use.  With a stage barrier (SYNC 0 / 1) every buffer is free as
  // soon as the barrier is passed: DIST = NBUF - 1.  With flags (SYNC 2) the spare buffers are SLACK: DIST < NBUF - 1 lets
  // a wave run up to NBUF - 1 - DIST stages ahead of the slowest sibling before it has to wait for a free buffer.
  static_assert(DIST >= 1 && DIST <= NBUF - 1 && (SYNC == 2 || DIST == NBUF - 1), "");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wbuf = reinterpret_cast<unsigned short*>(smem_raw);
  // SYNC 2: flags[0][w] = weight stages whose pieces issued by wave w have landed, flags[1][w] = stages wave w has finished
  // reading (monotone counters, one writer each)
  volatile int* flags = reinterpret_cast<volatile int*>(smem_raw + NBUF * BUF * 2);

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  {      // XCD-contiguous tile ranges (cdna_hip_programming.md T1, bijective)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int tile0 = (bid * WAVES + wave) * TPW;
  (void)n_tiles;
  const int loff_b = lane * 16;

  if constexpr (SYNC == 2) {
    if (tid < 2 * WAVES) flags[tid] = 0;
  }

  // ---- weight stage requests (the production mapping: edge_layer_kernel.h FUSED_DMA_PIECE) ----
  unsigned dvoff;
  {
    const int entry0 = (PP * wave) * 32 + (lane >> 1), half = (lane & 1) ^ ((lane >> 4) & 1);
    dvoff = (entry0 >> 8) * 4096 + (entry0 & 255) * 16 + half * 8;
  }
  const __amdgpu_buffer_rsrc_t rs_c =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c_planes), 0, 0x7fffffff, 0x00020000);
  const int plane_bytes = (int)plane_stride * 2;
#define LAB_DMA_STAGE(t)                                                                                              \
  {                                                                                                                   \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) _Pragma("unroll") for (int i = 0; i < PP; ++i)                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                     \
            rs_c, (__attribute__((address_space(3))) void*)(wbuf + ((t) % NBUF) * BUF + pl * PLANE + (PP * wave + i) * 512), \
            16, dvoff * 2, (t)*4096 * 2 + pl * plane_bytes + i * 512 * 2, 0, 0);                                      \
  }

  // ---- e stream: register ring, RING slabs per tile ----
  v4f er[TPW][RING][2];
  __amdgpu_buffer_rsrc_t rs_e[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u)
    rs_e[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e) + (long long)(tile0 + u) * (32 * H), 0, 32 * H * 4,
                                                0x00020000);
#define LAB_E_LOAD(u, ks)                                                                                           \
  {                                                                                                                 \
    er[u][(ks) % RING][0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_e[u], loff_b, (ks)*2048, 0)); \
    er[u][(ks) % RING][1] =                                                                                         \
        __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_e[u], loff_b, (ks)*2048 + 1024, 0));      \
  }
#pragma unroll
  for (int d = 0; d < RING; ++d)
#pragma unroll
    for (int u = 0; u < TPW; ++u) LAB_E_LOAD(u, d)
#pragma unroll
  for (int s = 0; s < DIST; ++s) LAB_DMA_STAGE(s)

  v16f acc[TPW][8];
#pragma unroll
  for (int u = 0; u < TPW; ++u)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][nb][r] = 0.0f;

  const int a_off = wslot(l31, hh);

  // synthetic epilogue (VMIX): element = {fma, exp2, add, rcp, mul, fma, max, add} twice; LDS round trips on the wave's scratch
  float* lscr = reinterpret_cast<float*>(smem_raw + NBUF * BUF * 2 + 256) + wave * 64 * 20 + lane * 4;
  const float ca = 1.0009765625f, cb = 0.03125f, cc = -1.4426950408889634f, cg = 0.998046875f, ct = 0.015625f;
#define LAB_ELEM(x)                                                                                  \
  {                                                                                                  \
    _Pragma("unroll") for (int rnd_ = 0; rnd_ < 2; ++rnd_) {                                         \
      const float z_ = __builtin_fmaf((x), ca, cb);                                                  \
      const float s_ = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z_ * cc));                \
      float y_ = __builtin_fmaf(z_ * s_, cg, cb);                                                    \
      (x) = __builtin_fmaxf(y_, 0.0f) + ct;                                                          \
    }                                                                                                \
  }
  // chunk c (0..63) of an epilogue over the 128 values v[0..127]: two elements + 4 LDS operations
#define LAB_CHUNK(v, c)                                                                              \
  {                                                                                                  \
    LAB_ELEM(v[2 * (c)])                                                                             \
    LAB_ELEM(v[2 * (c) + 1])                                                                         \
    *reinterpret_cast<v4f*>(lscr + ((c) & 3) * 256 * 0) = v4f{v[2 * (c)], v[2 * (c) + 1], 0.f, 0.f}; \
    const v4f r_ = *reinterpret_cast<const v4f*>(lscr);                                              \
    v[2 * (c)] += r_[2];                                                                             \
    *reinterpret_cast<v4f*>(lscr) = v4f{v[2 * (c) + 1], v[2 * (c)], 0.f, 0.f};                       \
    const v4f q_ = *reinterpret_cast<const v4f*>(lscr);                                              \
    v[2 * (c) + 1] += q_[3];                                                                         \
  }
  float ex[VMIX == 2 ? 128 : 1];
  if constexpr (VMIX == 2) {
#pragma unroll
    for (int i = 0; i < 128; ++i) ex[i] = er[0][0][0][i & 3] * (float)(i + 1);
  }

  // first stage landed (every wave's pieces)
  if constexpr (SYNC == 2) {
    wait_vmcnt<(DIST - 1) * PPW>();
    __builtin_amdgcn_s_barrier();      // (the flag words were zeroed above; one barrier per tile, none per stage)
  } else if constexpr (SYNC == 1) {
    wait_vmcnt<(DIST - 1) * PPW>();
    __builtin_amdgcn_s_barrier();
  } else {
    wait_vmcnt<0>();
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < NS; ++t) {
    // B operands of slab t
    frag xh[TPW], xl[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const v4f c0 = er[u][t % RING][0], c1 = er[u][t % RING][1];
      const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      split8<T>(xs, xh[u], xl[u]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SYNC == 2) {
      // buffer (t + DIST) % NBUF held stage t + DIST - NBUF: every wave must have finished reading it
      if (t + DIST - NBUF >= 0 && t + DIST < NS) lab_flags_wait<WAVES>(flags + WAVES, t + DIST - NBUF + 1);
    }
    if (t + DIST < NS) LAB_DMA_STAGE(t + DIST)
    if (t + RING < NS) {
#pragma unroll
      for (int u = 0; u < TPW; ++u) LAB_E_LOAD(u, t + RING)
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SYNC == 2) {
      if (t >= 1) {      // stage t: every wave's pieces landed?  (stage 0 was met by the barrier above)
        lab_flags_wait<WAVES>(flags, t);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned short* wb = wbuf + (t % NBUF) * BUF + a_off;
    frag fh[4], fl[4];
#define LAB_FRAG(bi, slot)                                                      \
  {                                                                             \
    fh[slot] = *reinterpret_cast<const frag*>(wb + (bi)*32 * 16);               \
    fl[slot] = *reinterpret_cast<const frag*>(wb + PLANE + (bi)*32 * 16);       \
  }
    LAB_FRAG(0, 0)
    LAB_FRAG(1, 1)
#pragma unroll
    for (int bp = 0; bp < 4; ++bp) {
      if (bp + 1 < 4) {
        LAB_FRAG(2 * bp + 2, 2 * ((bp + 1) & 1))
        LAB_FRAG(2 * bp + 3, 2 * ((bp + 1) & 1) + 1)
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(1);
      const int s0 = 2 * (bp & 1), s1 = s0 + 1, n0 = 2 * bp, n1 = n0 + 1;
      if constexpr (VMIX == 4) {
        // ROUND 6, VERDICT r5 #7 (TIMING / POWER ONLY - the results are not the product): the two correction products of a block
        // (hi.lo, lo.hi) as ONE v_mfma_scale_f32_32x32x64_f8f6f4 per TWO k slabs - K = 64 bytes = [lo | hi] x [hi ; lo] of 32 real k,
        // 16 passes against the 4 x 8 passes of the four fp16 correction MFMAs it stands for.  Issued on the odd stages; its operands
        // are the raw bytes of this stage's fp16 fragments (realistic bit activity, meaningless values).
        if (t & 1) {
          typedef int v8i_ __attribute__((ext_vector_type(8)));
          typedef int v4i_ __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int u = 0; u < TPW; ++u) {
            const v4i_ xa = __builtin_bit_cast(v4i_, xh[u]), xb = __builtin_bit_cast(v4i_, xl[u]);
            const v8i_ bx = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
            const v4i_ a0 = __builtin_bit_cast(v4i_, fl[s0]), a1 = __builtin_bit_cast(v4i_, fh[s0]);
            const v4i_ b0 = __builtin_bit_cast(v4i_, fl[s1]), b1 = __builtin_bit_cast(v4i_, fh[s1]);
            const v8i_ w0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const v8i_ w1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            acc[u][n0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w0, bx, acc[u][n0], 0, 0, 0, 127, 0, 127);
            acc[u][n1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w1, bx, acc[u][n1], 0, 0, 0, 127, 0, 127);
          }
        }
      } else {
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        acc[u][n0] = T::mfma(fl[s0], xh[u], acc[u][n0]);
        acc[u][n1] = T::mfma(fl[s1], xh[u], acc[u][n1]);
      }
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        acc[u][n0] = T::mfma(fh[s0], xl[u], acc[u][n0]);
        acc[u][n1] = T::mfma(fh[s1], xl[u], acc[u][n1]);
      }
      }
#pragma unroll
      for (int u = 0; u < TPW; ++u) {
        acc[u][n0] = T::mfma(fh[s0], xh[u], acc[u][n0]);
        acc[u][n1] = T::mfma(fh[s1], xh[u], acc[u][n1]);
      }
      if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(0);
      if constexpr (VMIX == 2) { LAB_CHUNK(ex, 4 * t + bp) }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef LAB_FRAG
    if (t + 1 < NS) {
      if constexpr (SYNC == 0) {
        if (t + RING < NS) wait_vmcnt<2 * TPW>(); else wait_vmcnt<0>();
        __syncthreads();
      } else {
        constexpr int kTab[NS] = {
            allowed_in_flight<TPW, PPW, DIST, RING>(0),  allowed_in_flight<TPW, PPW, DIST, RING>(1),
            allowed_in_flight<TPW, PPW, DIST, RING>(2),  allowed_in_flight<TPW, PPW, DIST, RING>(3),
            allowed_in_flight<TPW, PPW, DIST, RING>(4),  allowed_in_flight<TPW, PPW, DIST, RING>(5),
            allowed_in_flight<TPW, PPW, DIST, RING>(6),  allowed_in_flight<TPW, PPW, DIST, RING>(7),
            allowed_in_flight<TPW, PPW, DIST, RING>(8),  allowed_in_flight<TPW, PPW, DIST, RING>(9),
            allowed_in_flight<TPW, PPW, DIST, RING>(10), allowed_in_flight<TPW, PPW, DIST, RING>(11),
            allowed_in_flight<TPW, PPW, DIST, RING>(12), allowed_in_flight<TPW, PPW, DIST, RING>(13),
            allowed_in_flight<TPW, PPW, DIST, RING>(14), 0};
        // (switch on the unrolled t: the wait needs an immediate)
        switch (kTab[t]) {
#define LAB_W(n) case n: wait_vmcnt<n>(); break;
          LAB_W(0) LAB_W(1) LAB_W(2) LAB_W(3) LAB_W(4) LAB_W(5) LAB_W(6) LAB_W(7) LAB_W(8) LAB_W(9) LAB_W(10) LAB_W(11)
          LAB_W(12) LAB_W(13) LAB_W(14) LAB_W(15) LAB_W(16) LAB_W(17) LAB_W(18) LAB_W(19) LAB_W(20) LAB_W(21) LAB_W(22)
          LAB_W(23) LAB_W(24) LAB_W(25) LAB_W(26) LAB_W(27) LAB_W(28) LAB_W(29) LAB_W(30) LAB_W(31) LAB_W(32)
#undef LAB_W
          default: wait_vmcnt<0>(); break;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SYNC == 1) {
          __builtin_amdgcn_s_barrier();
        } else {
          // publish: my pieces of stage t+1 have landed, I have finished reading stage t
          if (lane == 0) {
            flags[wave] = t + 1;
            flags[WAVES + wave] = t + 1;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef LAB_DMA_STAGE
#undef LAB_E_LOAD
  if constexpr (VMIX == 1) {      // the tile's own accumulators: 128 values per lane
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      float* v = reinterpret_cast<float*>(&acc[0][c >> 3]) + 2 * (c & 7) - 2 * c;      // v[2c], v[2c+1] = the pair c of block c / 8
      LAB_CHUNK(v, c)
    }
  }
  if constexpr (VMIX == 2) {      // keep the interleaved epilogue alive
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 128; ++i) sum += ex[i];
    if (sum == 123456.789f) out[0] = sum;
  }
#undef LAB_CHUNK
#undef LAB_ELEM

  // out (tiled like e) = acc / 2^kc
  if (do_store) {
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      float* ot = out + (long long)(tile0 + u) * (32 * H) + lane * 4;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v4f v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = acc[u][nb][4 * g + q] * inv_c;
          __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(ot + (2 * nb + (g >> 1)) * 512 + (g & 1) * 256));
        }
    }
  } else {
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the "v" constraint does not exist on the host pass: clang then drops the whole
        asm volatile("" ::"v"(acc[u][nb]));      //  instantiation - stub included - without a diagnostic)
#endif
      }
  }
}

template <int EPW, int WAVES, int NBUF, int SYNC, int RING, int PRIO, int MINB, int DIST = NBUF - 1, int VMIX = 0>
hipError_t launch(const float* e, const unsigned short* planes, float* out, int n_edges, float inv_c, int do_store,
                  int lds_pad, hipStream_t st) {
  static std::atomic<unsigned long long> attr_devices{0};
  hipError_t er = ensure_max_dynamic_lds(
      attr_devices, reinterpret_cast<const void*>(&stage_lab_kernel<EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB, DIST, VMIX>), 160 * 1024);
  if (er != hipSuccess) return er;
  const int per_wg = EPW * WAVES;
  if (n_edges % per_wg != 0) return hipErrorInvalidValue;
  const int lds = NBUF * BUF * 2 + 256 + WAVES * 64 * 20 * 4 + lds_pad;      // stage buffers | flags | VMIX scratch
  hipLaunchKernelGGL((stage_lab_kernel<EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB, DIST, VMIX>), dim3((unsigned)(n_edges / per_wg)),
                     dim3(64 * WAVES), lds, st, e, planes, (long long)H * H, out, n_edges / 32, inv_c, do_store);
  return hipGetLastError();
}


// ---- the same GEMM on the 16x16x32 MFMA shape -----------------------------------------------------------------------
// v_mfma_f32_16x16x32_f16 has the rate of v_mfma_f32_32x32x16_f16 (16 against 32 cycles for half the multiply-adds) but moves
// fewer accumulator registers per multiply-add: C in + D out are 4 + 4 registers for 8,192 multiply-adds per lane group against
// 16 + 16 for 16,384 - half - while the A / B operand registers per multiply-add double; LDS traffic is the same (an A fragment is
// 512 weights in both shapes).  On a chip that runs this GEMM on its power cap (profiles/r04/lab_power_randn_vs_zeros.txt) the
// question is what the total is worth.  Geometry and synchronisation = the production scheme (32 edges per wave, 4 waves, two
// workgroups per CU, double-buffered 16 KiB stages, vmcnt + __syncthreads()).  A stage holds 128 weight rows x 32 k (two k slabs
// of the production plane layout), i.e. stage t = (k block t / 2, row half t % 2); a wave owns 16 feature tiles x 2 edge halves of
// 16 x 16 accumulators (128 registers, as before).  Operand k order inside an instruction: lane group q = 2 slab + half holds the
// eight k of that half of that slab, A and B alike.  Results are real (checked against float64 by the bench script).
template <int MINB>
__global__ __launch_bounds__(256, MINB) void stage_lab16_kernel(const float* __restrict__ e,
                                                                const unsigned short* __restrict__ c_planes, long long plane_stride,
                                                                float* __restrict__ out, float inv_c, int do_store) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef FFp16 T;
  constexpr int WAVES = 4, PP = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* wbuf = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, idx = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int tile = bid * WAVES + wave;
  const int c16 = lane & 15, q4 = lane >> 4, sl = q4 >> 1, hq = q4 & 1;

  // weight stage requests: piece p = PP wave + i of a plane = (slab p / 4, rows 32 (p % 4) .. + 31 of the row half); lane L fills
  // LDS slot (row L / 2, half slot L % 2) with the half (L % 2) ^ (row / 8 % 2) of that row (the production swizzle)
  const __amdgpu_buffer_rsrc_t rs_c =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c_planes), 0, 0x7fffffff, 0x00020000);
  const int plane_bytes = (int)plane_stride * 2;
  unsigned dvoff[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int p = PP * wave + i, half = (lane & 1) ^ ((lane >> 4) & 1);
    dvoff[i] = ((p >> 2) * 4096 + (32 * (p & 3) + (lane >> 1)) * 16 + half * 8) * 2;      // bytes; + (2 kb) slabs + row half
  }
#define LAB16_DMA_STAGE(t)                                                                                                 \
  {                                                                                                                        \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) _Pragma("unroll") for (int i = 0; i < PP; ++i)                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                          \
            rs_c, (__attribute__((address_space(3))) void*)(wbuf + ((t)&1) * BUF + pl * PLANE + (PP * wave + i) * 512),    \
            16, dvoff[i], (((t) >> 1) * 2 * 4096 + ((t)&1) * 128 * 16) * 2 + pl * plane_bytes, 0, 0);                      \
  }
  // e stream: per k block and edge half two float4 per lane (tiled layout: slab 2 kb + sl, i, lane slot = edge + 32 half)
  const __amdgpu_buffer_rsrc_t rs_e =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e) + (long long)tile * (32 * H), 0, 32 * H * 4, 0x00020000);
  const int e_voff = (sl * 512 + (c16 + 32 * hq) * 4) * 4;      // bytes; + eh * 16 lanes * 16 B, + kb * 2 slabs, + i * 1 KiB
  v4f er[2][2][2];
#define LAB16_E_LOAD(kb)                                                                                                   \
  {                                                                                                                        \
    _Pragma("unroll") for (int eh = 0; eh < 2; ++eh) _Pragma("unroll") for (int i = 0; i < 2; ++i)                         \
        er[(kb)&1][eh][i] = __builtin_bit_cast(                                                                            \
            v4f, __builtin_amdgcn_raw_buffer_load_b128(rs_e, e_voff + eh * 256, (kb)*4096 + i * 1024, 0));                 \
  }
  LAB16_E_LOAD(0)
  LAB16_E_LOAD(1)
  LAB16_DMA_STAGE(0)

  v4f acc[16][2];
#pragma unroll
  for (int ft = 0; ft < 16; ++ft)
#pragma unroll
    for (int eh = 0; eh < 2; ++eh) acc[ft][eh] = v4f{0.f, 0.f, 0.f, 0.f};

  // A fragment of feature tile ft8 of the stage: row 16 ft8 + c16, slab sl, half hq (swizzled)
  const int a_off = sl * 2048 + c16 * 16 + ((hq ^ ((c16 >> 3) & 1)) << 3);
  wait_vmcnt<0>();
  __syncthreads();

  h8 xh[2], xl[2];
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    const int kb = t >> 1, rh = t & 1;
    if (rh == 0) {      // B operands of k block kb, both edge halves
#pragma unroll
      for (int eh = 0; eh < 2; ++eh) {
        const v4f c0 = er[kb & 1][eh][0], c1 = er[kb & 1][eh][1];
        const float xs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        split8<T>(xs, xh[eh], xl[eh]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < NS) LAB16_DMA_STAGE(t + 1)
    if (rh == 0 && kb + 2 < 8) LAB16_E_LOAD(kb + 2)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned short* wb = wbuf + (t & 1) * BUF + a_off;
    h8 fh[2], fl[2];
    fh[0] = *reinterpret_cast<const h8*>(wb);
    fl[0] = *reinterpret_cast<const h8*>(wb + PLANE);
#pragma unroll
    for (int f8 = 0; f8 < 8; ++f8) {
      if (f8 + 1 < 8) {
        fh[(f8 + 1) & 1] = *reinterpret_cast<const h8*>(wb + (f8 + 1) * 256);
        fl[(f8 + 1) & 1] = *reinterpret_cast<const h8*>(wb + PLANE + (f8 + 1) * 256);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ft = 8 * rh + f8, s0 = f8 & 1;
      acc[ft][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[s0], xh[0], acc[ft][0], 0, 0, 0);
      acc[ft][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[s0], xh[1], acc[ft][1], 0, 0, 0);
      acc[ft][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[s0], xl[0], acc[ft][0], 0, 0, 0);
      acc[ft][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[s0], xl[1], acc[ft][1], 0, 0, 0);
      acc[ft][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[s0], xh[0], acc[ft][0], 0, 0, 0);
      acc[ft][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[s0], xh[1], acc[ft][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + 1 < NS) {
      if (rh == 0 && kb + 2 < 8) wait_vmcnt<4>(); else wait_vmcnt<0>();      // the e loads of this stage stay in flight
      __syncthreads();
    }
  }
#undef LAB16_DMA_STAGE
#undef LAB16_E_LOAD
  if (do_store) {      // out tiled like e: feature f = 16 ft + 4 q4 + 0..3 of edge 16 eh + c16
    float* ot = out + (long long)tile * (32 * H);
#pragma unroll
    for (int ft = 0; ft < 16; ++ft)
#pragma unroll
      for (int eh = 0; eh < 2; ++eh) {
        const v4f v = acc[ft][eh] * inv_c;
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(ot + ft * 512 + (q4 >> 1) * 256 + (16 * eh + c16 + 32 * (q4 & 1)) * 4));
      }
  } else {
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::"v"(acc[ft][0]), "v"(acc[ft][1]));
#endif
    }
  }
}

template <int MINB>
hipError_t launch16(const float* e, const unsigned short* planes, float* out, int n_edges, float inv_c, int do_store, int lds_pad,
                    hipStream_t st) {
  static std::atomic<unsigned long long> attr_devices{0};
  hipError_t er = ensure_max_dynamic_lds(attr_devices, reinterpret_cast<const void*>(&stage_lab16_kernel<MINB>), 160 * 1024);
  if (er != hipSuccess) return er;
  if (n_edges % 128 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL((stage_lab16_kernel<MINB>), dim3((unsigned)(n_edges / 128)), dim3(256), 2 * BUF * 2 + lds_pad, st, e, planes,
                     (long long)H * H, out, inv_c, do_store);
  return hipGetLastError();
}

}  // namespace LAB_NS
}  // namespace difusco

#ifndef DIFUSCO_LAB_NOPK
namespace difusco {
namespace lab {
// Re-read probe (where is the residual re-read of e served?): one pass over `n4` float4 with the cache policy of
// the fused kernel's e accesses (aux 2 = non-temporal, 0 = default); the sum keeps the loads alive.  Timing pass 2 over a buffer
// against pass 1 (cold) for buffer sizes around the re-use distance of the kernel tells whether the memory-side cache (MALL,
// 256 MiB) serves a second read of lines that were streamed through the L2 non-temporally a few tens of microseconds earlier.
template <int AUX>
__global__ __launch_bounds__(256) void reread_probe_kernel(const float* __restrict__ buf, long long n4, float* __restrict__ sink) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, 0x7fffffff, 0x00020000);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long base = (long long)blockIdx.x * blockDim.x; base < n4; base += stride) {
    const long long i = base + threadIdx.x;
    if (i < n4) {
      const float* p = buf + i * 4;
      v4f v;
      if constexpr (AUX == 2) v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
      else v = *reinterpret_cast<const v4f*>(p);
      acc += v;
    }
  }
  (void)rs;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123456.789f) sink[0] = acc[0];
}
}  // namespace lab
}  // namespace difusco
extern "C" int difusco_lab_reread_pass(const float* buf, long long n_floats, int nontemporal, float* sink, void* stream) {
  using namespace difusco::lab;
  const long long n4 = n_floats / 4;
  if (nontemporal) hipLaunchKernelGGL((reread_probe_kernel<2>), dim3(2048), dim3(256), 0, (hipStream_t)stream, buf, n4, sink);
  else hipLaunchKernelGGL((reread_probe_kernel<0>), dim3(2048), dim3(256), 0, (hipStream_t)stream, buf, n4, sink);
  return hipGetLastError() == hipSuccess ? DIFUSCO_OK : DIFUSCO_EHIP;
}

// ROUND 6 probe (scripts/lab/r06/f8_probe.py): ONE wave, one v_mfma_scale_f32_32x32x64_f8f6f4 (both operands E4M3) on caller-supplied
// register images - a[lane][8 dwords], b[lane][8 dwords], one E8M0 scale byte per lane and operand - and, in the same launch, one
// v_permlane32_swap of two dwords per lane: the operand / scale / result layouts a split-precision kernel with FP8 correction products
// would have to be written for.  out: [lane][16] accumulators, then [lane][2] swapped dwords.
namespace difusco {
__global__ void lab_f8_probe_kernel(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ sa,
                                    const int* __restrict__ sb, const unsigned* __restrict__ sw, float* __restrict__ out,
                                    const float* __restrict__ cin) {
  typedef int v8i_ __attribute__((ext_vector_type(8)));
  typedef float v16f_ __attribute__((ext_vector_type(16)));
  const int lane = threadIdx.x;
  v8i_ av, bv;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    av[j] = a[lane * 8 + j];
    bv[j] = b[lane * 8 + j];
  }
  v16f_ acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = cin != nullptr ? cin[lane * 16 + r] : 0.0f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa[lane], 0, sb[lane]);
#pragma unroll
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
  // (inline assembly: hipcc 7.2 lowers __builtin_amdgcn_permlane32_swap to the instruction but hands back its first result twice)
  unsigned p0 = sw[lane * 2], p1 = sw[lane * 2 + 1];
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p0), "+v"(p1));
  out[64 * 16 + lane * 2] = __builtin_bit_cast(float, p0);
  out[64 * 16 + lane * 2 + 1] = __builtin_bit_cast(float, p1);
  if (lane == 0) {      // v_cvt_scalef32_pk_fp8_f32: is the result E4M3(src / scale) or E4M3(src * scale)?  (100, -3) with scale 256 and 1/16
    typedef short v2s_ __attribute__((ext_vector_type(2)));
    v2s_ w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, 100.0f, -3.0f, 256.0f, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, 100.0f, -3.0f, 0.0625f, true);
    out[64 * 16 + 128] = __builtin_bit_cast(float, w);
  }
}
}  // namespace difusco
extern "C" int difusco_lab_f8_probe(const int* a, const int* b, const int* sa, const int* sb, const unsigned* sw, float* out, void* stream,
                                    const float* cin) {
  hipLaunchKernelGGL(difusco::lab_f8_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, sa, sb, sw, out, cin);
  return hipGetLastError() == hipSuccess ? DIFUSCO_OK : DIFUSCO_EHIP;
}
#endif

extern "C" {
// variant = EPW/32 * 100000 + WAVES * 10000 + NBUF * 1000 + SYNC * 100 + RING * 10 + PRIO; MINB follows from the geometry
// (64-edge waves and 8-wave workgroups are one workgroup per CU, the production geometry two).  planes: the fp16 hi | lo
// planes of C (weights.split_planes output + 3 H H 16-bit elements); e / out tiled [n_edges, 256]; n_edges a multiple of the
// edges per workgroup.  lds_pad: extra dynamic LDS bytes (to pin the number of co-resident workgroups).
int LAB_ENTRY(int variant, const float* e, const void* planes, float* out, int n_edges, float inv_c, int do_store,
              int lds_pad, void* stream) {
  using namespace difusco::LAB_NS;
  const unsigned short* pl = reinterpret_cast<const unsigned short*>(planes);
  hipStream_t st = (hipStream_t)stream;
  hipError_t er = hipErrorInvalidValue;
#define LAB_CASE(code, EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB) \
  case code: er = launch<EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
#define LAB_CASE_D(code, EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB, DIST) \
  case code: er = launch<EPW, WAVES, NBUF, SYNC, RING, PRIO, MINB, DIST>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
  switch (variant) {
    // production geometry: 32 edges per wave, 4 waves, two workgroups per CU
    LAB_CASE(142020, 32, 4, 2, 0, 2, 0, 2)      // production scheme (double buffer, vmcnt + __syncthreads)
    LAB_CASE(142120, 32, 4, 2, 1, 2, 0, 2)      // raw barrier, counted waits, double buffer
    LAB_CASE(143120, 32, 4, 3, 1, 2, 0, 2)      // + third buffer: one stage in flight across the barrier
    LAB_CASE(143130, 32, 4, 3, 1, 3, 0, 2)      // + e ring of three slabs
    LAB_CASE(143121, 32, 4, 3, 1, 2, 1, 2)      // + s_setprio 1 around the MFMA groups
    LAB_CASE_D(143220, 32, 4, 3, 2, 2, 0, 2, 1)      // flags instead of the stage barrier: requests one stage ahead, one buffer of slack
    LAB_CASE_D(144220, 32, 4, 4, 2, 2, 0, 2, 2)      // flags, four buffers: two stages ahead, one buffer of slack
    // 64 edges per wave, 4 waves, ONE workgroup per CU (one wave per SIMD, up to 512 registers)
    LAB_CASE(242020, 64, 4, 2, 0, 2, 0, 1)
    LAB_CASE(243120, 64, 4, 3, 1, 2, 0, 1)
    LAB_CASE(244120, 64, 4, 4, 1, 2, 0, 1)
    LAB_CASE(244130, 64, 4, 4, 1, 3, 0, 1)
    LAB_CASE(244121, 64, 4, 4, 1, 2, 1, 1)
    LAB_CASE_D(244220, 64, 4, 4, 2, 2, 0, 1, 2)
    // 32 edges per wave, 8 waves in ONE workgroup per CU sharing the weight stream (two waves per SIMD in lock step)
    LAB_CASE(182020, 32, 8, 2, 0, 2, 0, 1)
    LAB_CASE(183120, 32, 8, 3, 1, 2, 0, 1)
    LAB_CASE(184120, 32, 8, 4, 1, 2, 0, 1)
    LAB_CASE(184121, 32, 8, 4, 1, 2, 1, 1)
    LAB_CASE_D(184220, 32, 8, 4, 2, 2, 0, 1, 2)
    // synthetic epilogue (VMIX): serial at two waves per SIMD | interleaved behind the MFMA groups at one wave per SIMD | serial at one wave per SIMD
    case 1142020: er = launch<32, 4, 2, 0, 2, 0, 2, 1, 1>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    case 2142020: er = launch<32, 4, 2, 0, 2, 0, 1, 1, 2>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    case 2143120: er = launch<32, 4, 3, 1, 2, 0, 1, 2, 2>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    case 3142020: er = launch<32, 4, 2, 0, 2, 0, 1, 1, 1>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    // round 6: hi.hi in fp16 + the two correction products as one FP8 MFMA per two slabs (timing / power only, production geometry)
    case 4142020: er = launch<32, 4, 2, 0, 2, 0, 2, 1, 4>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    // the 16x16x32 MFMA shape in the production scheme (two workgroups per CU | one)
    case 16142020: er = launch16<2>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    case 16142021: er = launch16<1>(e, pl, out, n_edges, inv_c, do_store, lds_pad, st); break;
    default: return difusco::set_error(DIFUSCO_EINVAL, "unknown lab variant %d", variant);
  }
#undef LAB_CASE
#undef LAB_CASE_D
  if (er != hipSuccess) return difusco::set_error(DIFUSCO_EHIP, "lab launch: %s", hipGetErrorString(er));
  return DIFUSCO_OK;
}
}
#endif  // DIFUSCO_PROFILING
