#!/bin/bash
# production 3955 (pair arithmetic incl. operand split and GEMM 2 output) vs 1907: bit comparison, parity, A/B
mkdir -p gpurun_out/p
timeout 200 python scripts/dev/fused_opt_diff.py 1907 3955 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider > gpurun_out/p/parity.log 2>&1; tail -1 gpurun_out/p/parity.log
for opt in 1907 3955 1907 3955; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > gpurun_out/p/bench_$opt.json
  python -c "import json; r=json.load(open('gpurun_out/p/bench_$opt.json')); print($opt, r['value'], r['ms_per_step'], r['roofline']['frac'])"
done
