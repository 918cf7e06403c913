#!/bin/bash
# which part of the buffer-resource e stream breaks parity?  (OPT bit 8 ring loads, 9 residual loads, 10 stores)
mkdir -p gpurun_out/m
for lib in default maxocc; do
  cp difusco_amd/lib/alt/$lib.so difusco_amd/lib/libdifusco_hip.so
  for opt in 115 371 627 1139 1907; do
    timeout 400 python - > gpurun_out/m/${lib}_$opt.log 2>&1 <<PY
import torch; torch.zeros(1, device="cuda")
from difusco_amd import _lib
_lib.check(_lib.lib().difusco_debug_set(7, $opt))
import pytest, sys
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", "test_edge_layer_fused or golden_h256_tsp_sparse or tsp1000_oracle", "-p", "no:cacheprovider"]))
PY
    echo "$lib $opt: $(tail -1 gpurun_out/m/${lib}_$opt.log)"
  done
done
