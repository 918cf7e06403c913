#!/bin/bash
# cache policy of the e accesses: 883 all non-temporal | +1024 GEMM 1 slabs default | +2048 residual default | +4096 stores default
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s; mkdir -p $O
for opt in 883 1907 2931 3955 4979 6003 883 1907 2931 3955 4979 6003; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --fused-opt $opt 2>/dev/null | grep '^{' > $O/bench_$opt.json
  python -c "import json; r=json.load(open('$O/bench_$opt.json')); print($opt, r['value'], r['ms_per_step'], r['roofline']['frac'])"
done
