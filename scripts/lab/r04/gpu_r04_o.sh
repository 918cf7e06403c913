#!/bin/bash
# Round 4, GPU call O: ABI 10 (difusco_step_args.aggregation: sum / mean / max) - the new aggregation tests, then the whole
# GPU suite and smoke on that tree, and the default bench (the sum path must be where it was).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04o
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_aggregation.py -q -s --maxfail=10 > $OUT/pytest_agg.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_agg.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --deselect tests/test_gpu_aggregation.py > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python bench.py --no-workloads --cpu-steps 0 --no-exact-fp32 > $OUT/bench.json 2> $OUT/bench.err
grep -E "L_inf|passed|failed|error|exit" $OUT/pytest_agg.log | tail -50
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r04o/bench.json").read().strip().splitlines()[-1]); r = o["roofline"]
print("bench", round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "fused avg ms", round(r["avg_launch_ms"], 4), "other", round(r["other_ms_per_step"], 3))
PY
