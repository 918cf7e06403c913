// bf16-plane instantiations of the fused edge-layer kernel (edge_layer_kernel.h): precision DIFUSCO_PREC_BF16X3.
#include "edge_layer_kernel.h"

namespace difusco {
hipError_t launch_fused_bf16(int kind, FUSED_KIND_PARAMS) { return launch_fused_kind<FBf16>(kind, FUSED_KIND_ARGS); }
}  // namespace difusco
