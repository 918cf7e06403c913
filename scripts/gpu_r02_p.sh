#!/bin/bash
# issue priority (s_setprio) outside the GEMM phases: bit 9 gathers..LayerNorms, bit 10 prologue, bit 11 output phases
mkdir -p gpurun_out/p
for opt in 371 883 1907 2931 3955 371 883 1907 2931 3955; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > gpurun_out/p/bench_$opt.json
  python -c "import json; r=json.load(open('gpurun_out/p/bench_$opt.json')); print($opt, r['value'], r['ms_per_step'], r['roofline']['frac'])"
done
