#!/bin/bash
# OPT bit 9 (LDS-DMA pieces issued between the MFMA groups): parity, then A/B against production
mkdir -p gpurun_out/p
timeout 400 python - > gpurun_out/p/parity_883.log 2>&1 <<PY
import torch; torch.zeros(1, device="cuda")
from difusco_amd import _lib
_lib.check(_lib.lib().difusco_debug_set(7, 883))
import pytest, sys
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", "test_edge_layer_fused or golden_h256 or tsp1000_oracle", "-p", "no:cacheprovider"]))
PY
echo "parity 883: $(tail -1 gpurun_out/p/parity_883.log)"
for opt in 371 883 371 883; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > gpurun_out/p/bench_$opt.json
  python -c "import json; r=json.load(open('gpurun_out/p/bench_$opt.json')); print($opt, r['value'], r['ms_per_step'], r['roofline']['frac'])"
done
