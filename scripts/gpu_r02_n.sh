#!/bin/bash
# fused kernel: parity of every buffer-e variant (store hazard fixed), then timing under both schedulers
mkdir -p gpurun_out/n
for lib in default maxocc; do
  cp difusco_amd/lib/alt/$lib.so difusco_amd/lib/libdifusco_hip.so
  for opt in 1139 1907; do
    timeout 400 python - > gpurun_out/n/${lib}_$opt.log 2>&1 <<PY
import torch; torch.zeros(1, device="cuda")
from difusco_amd import _lib
_lib.check(_lib.lib().difusco_debug_set(7, $opt))
import pytest, sys
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-q", "-x", "-m", "gpu", "-k", "test_edge_layer_fused or golden_h256_tsp_sparse or tsp1000_oracle", "-p", "no:cacheprovider"]))
PY
    echo "$lib $opt: $(tail -1 gpurun_out/n/${lib}_$opt.log)"
  done
done
run() { # name lib opt
  cp difusco_amd/lib/alt/$2.so difusco_amd/lib/libdifusco_hip.so
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --fused-opt $3 2>/dev/null | grep '^{' > gpurun_out/n/bench_$1.json
  python - <<PY
import json; r=json.load(open("gpurun_out/n/bench_$1.json")); print("$1", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["cpu_baseline"].get("parity_linf"))
PY
}
run default_115 default 115
run maxocc_627 maxocc 627
run maxocc_1907 maxocc 1907
run default_1907 default 1907
run maxocc_371 maxocc 371
run maxocc_1907b maxocc 1907
run default_115b default 115
run maxocc_627b maxocc 627
