#!/bin/bash
# node-row linear: lookahead depth + XCD-aware block order, A/B + parity
mkdir -p gpurun_out/k
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/k/pytest.log 2>&1; tail -3 gpurun_out/k/pytest.log
for d in 4 1 4 1; do
  python bench.py --steps 30 --warmup 5 --no-exact-fp32 --node-linear-depth $d 2>/dev/null | grep '^{' > gpurun_out/k/bench_d$d.json
  python - <<PY
import json; r=json.load(open("gpurun_out/k/bench_d$d.json")); print("depth $d", r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/k/prof -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-exact-fp32 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/k/prof -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-160
