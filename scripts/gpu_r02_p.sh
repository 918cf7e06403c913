#!/bin/bash
# OPT bit 11 (L2 touch of the next tile): parity, then A/B against production
mkdir -p gpurun_out/p
timeout 400 python - > gpurun_out/p/parity.log 2>&1 <<PY
import torch; torch.zeros(1, device="cuda")
from difusco_amd import _lib
_lib.check(_lib.lib().difusco_debug_set(7, 3955))
import pytest, sys
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", "test_edge_layer_fused or golden_h256 or tsp1000_oracle or tsp500_x16", "-p", "no:cacheprovider"]))
PY
echo "parity 3955: $(tail -1 gpurun_out/p/parity.log)"
for opt in 1907 3955 1907 3955; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > gpurun_out/p/bench_$opt.json
  python -c "import json; r=json.load(open('gpurun_out/p/bench_$opt.json')); print($opt, r['value'], r['ms_per_step'], r['roofline']['frac'])"
done
for opt in 1907 3955; do
  timeout 300 python bench.py --workload mis --steps 10 --warmup 2 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > gpurun_out/p/bench_mis_$opt.json
  python -c "import json; r=json.load(open('gpurun_out/p/bench_mis_$opt.json')); print('mis', $opt, r['value'], r['ms_per_step'])"
done
