#!/bin/bash
# bench.py over the four single-GPU workloads of BASELINE.json's configs (outputs under gpurun_out/)
mkdir -p gpurun_out
for w in tsp500 tsp1000 tsp10000 mis; do
  timeout 600 python bench.py --workload $w --steps ${STEPS:-10} --warmup 2 --cpu-steps ${CPU_STEPS:-0} --profile-all > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  echo "== $w rc=$?"; tail -c 600 gpurun_out/bench_$w.err | tail -3
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$w.json"))
    r = d.get("roofline", {})
    print("$w", round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step | roofline", r.get("bound"), round(r.get("frac", 0), 3), "| kernels ms/step", round(d.get("kernels", {}).get("sum_ms_per_step", 0), 3), "| E", d["config"]["edges_rank0"], "N", d["config"]["nodes_rank0"])
except Exception as ex:
    print("$w failed", ex)
PY
done
