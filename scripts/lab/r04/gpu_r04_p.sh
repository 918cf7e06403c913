#!/bin/bash
# Round 4, GPU call P: aggregation = max on the FUSED layers (OPT bit 19 instantiations): the aggregation tests, the parity
# tests of the fused path (the production instantiations are byte-identical, re-checked anyway), and sum / mean / max side by side
# on the default workload (fused and unfused for max).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_aggregation.py -q -s --maxfail=10 > $OUT/pytest_agg.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_agg.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q --maxfail=10 -k "golden or oracle or dense or bench_workload or prepared or fixture" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_subset.log
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for agg in sum mean max; do
  timeout 300 python bench.py $AB --aggregation $agg > $OUT/bench_$agg.json 2> $OUT/bench_$agg.err
done
timeout 300 python bench.py $AB --aggregation max --no-fusion > $OUT/bench_max_unfused.json 2> $OUT/bench_max_unfused.err
timeout 300 python bench.py $AB --aggregation max --workload mis > $OUT/bench_max_mis.json 2> $OUT/bench_max_mis.err
timeout 300 python bench.py $AB --aggregation sum --workload mis > $OUT/bench_sum_mis.json 2> $OUT/bench_sum_mis.err
grep -E "passed|failed|error|exit" $OUT/pytest_agg.log | tail -5
tail -2 $OUT/pytest_subset.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04p/bench_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1]); r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "dominant avg ms", round(r.get("avg_launch_ms", 0), 4))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json", ".err")).read()[-400:])
PY
