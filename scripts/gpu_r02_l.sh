#!/bin/bash
# fused kernel: e stream through a buffer resource (OPT bit 8) and the iterative-maxocc scheduler (no scratch), A/B
mkdir -p gpurun_out/l
run() { # name lib opt
  cp difusco_amd/lib/alt/$2.so difusco_amd/lib/libdifusco_hip.so
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --fused-opt $3 2>/dev/null | grep '^{' > gpurun_out/l/bench_$1.json
  python - <<PY
import json; r=json.load(open("gpurun_out/l/bench_$1.json")); print("$1", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["cpu_baseline"].get("parity_linf"))
PY
}
run default_115 default 115
run default_371 default 371
run maxocc_115 maxocc 115
run maxocc_371 maxocc 371
run default_115b default 115
run maxocc_371b maxocc 371
run default_371b default 371
run maxocc_115b maxocc 115
