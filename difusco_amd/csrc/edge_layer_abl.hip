// Profiling-only variants of the fp16 middle-layer kernel (edge_layer_kernel.h): compile-time ablation masks selected
// with difusco_debug_set(0, mask).  Results are WRONG for every mask except 16 (production code + phase timestamps).
#include "edge_layer_kernel.h"

namespace difusco {
hipError_t launch_fused_ablation(int mask, FUSED_KIND_PARAMS) {
  (void)l0_table; (void)l0_x; (void)l0_perm; (void)gn_tile;
#define ABL_TAIL nullptr, nullptr, nullptr, nullptr, scales, etmax_in, etmax_out
#define ABL_ARGS e, node4, row, col, n_edges, c_planes, o_planes, plane_stride, b_c, g_e, b_e, tbias, g_o, b_o, b_out, \
                 time_on_edge, part, direct, stream
  switch (mask) {
    case 1: return launch_fused_t<FFp16, 1, FUSED_NW>(ABL_ARGS, ABL_TAIL);      // no neighbour-table gathers
    case 2: return launch_fused_t<FFp16, 2, FUSED_NW>(ABL_ARGS, ABL_TAIL);      // no neighbour sum
    case 4: return launch_fused_t<FFp16, 4, FUSED_NW>(ABL_ARGS, ABL_TAIL);      // no LayerNorm / activation math
    case 8: return launch_fused_t<FFp16, 8, FUSED_NW>(ABL_ARGS, ABL_TAIL);      // no GEMM 2
    case 15: return launch_fused_t<FFp16, 15, FUSED_NW>(ABL_ARGS, ABL_TAIL);    // GEMM 1 + weight streaming only
    case 16: return launch_fused_t<FFp16, 16, FUSED_NW>(ABL_ARGS, ABL_TAIL);    // production code + phase timestamps
    case 17: return launch_fused_t<FFp16, 15, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);   // 15 with the production options
    case 18: return launch_fused_t<FFp16, 16, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // 16 (phase stamps), production options (full-line gathers)
    case 19: return launch_fused_t<FFp16, 16, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);   // stamps, round 2's options (register gathers)
    case 34: return launch_fused_t<FFp16, 16 + 1024, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // stamps, gather requests never waited for
    case 35: return launch_fused_t<FFp16, 16 + 1024 + 2048, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // stamps, no gather requests, no waits
    case 36: return launch_fused_t<FFp16, 1024, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // (no stamps) gather requests never waited for
    case 37: return launch_fused_t<FFp16, 1024 + 2048, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // (no stamps) no gather requests, no waits
    // round 6, traffic attribution (profiles/r06/reread_attribution.txt): ONE access class off at a time, production options, no stamps
    case 38: return launch_fused_t<FFp16, 16384, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // no weight-stage refills / barriers (stale LDS)
    case 39: return launch_fused_t<FFp16, 32768, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // no e stream into GEMM 1
    case 40: return launch_fused_t<FFp16, 32, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);      // no residual read, no e store
    case 41: return launch_fused_t<FFp16, 128, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);     // no B h[i] loads
    case 42: return launch_fused_t<FFp16, 128 + 1024 + 2048, FUSED_NW, false, false, 0, FUSED_OPT>(ABL_ARGS, ABL_TAIL);   // no node-table access at all
    case 22: return launch_fused_t<FFp16, 16 + 32768, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);   // stamps, no e stream
    case 23: return launch_fused_t<FFp16, 16 + 16384, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);   // stamps, no stage refills / barriers
    case 24: return launch_fused_t<FFp16, 16 + 1, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);       // stamps, no gathers
    case 27: return launch_fused_t<FFp16, 16 + 65536, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);     // stamps, stage requests not waited for
    case 28: return launch_fused_t<FFp16, 16 + 131072, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);    // stamps, no stage barrier
    case 29: return launch_fused_t<FFp16, 16 + 196608, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);    // stamps, neither (requests still issued)
    case 30: return launch_fused_t<FFp16, 16 + 128, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);       // stamps, no B h[i] gathers
    case 31: return launch_fused_t<FFp16, 16 + 256, FUSED_NW, false, false, 0, FUSED_OPT_R2>(ABL_ARGS, ABL_TAIL);       // stamps, no A h[j] / V h[j] gathers
    case 32:      // stamps, LAST layer of a TSP step (no V gathers / gate / neighbour sum; GroupNorm partial sums go to `part`)
      return launch_fused_t<FFp16, 16, FUSED_NW, false, true, 1, FUSED_OPT_R2>(ABL_ARGS, nullptr, nullptr, nullptr, part, scales, etmax_in, nullptr);
    case 33:      // stamps, FIRST layer (two-row table instead of e and GEMM 1; the table rows are taken from b_c .. for timing only)
      return launch_fused_t<FFp16, 16, FUSED_NW, true, false, 0, FUSED_OPT_R2>(ABL_ARGS, node4, nullptr, nullptr, nullptr, scales, etmax_in, etmax_out);
    default: return hipErrorInvalidValue;
  }
#undef ABL_ARGS
#undef ABL_TAIL
}
}  // namespace difusco
