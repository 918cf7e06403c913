#!/bin/bash
# Round 4, GPU call U: the default bench line of the final tree (ABI 10) for profiles/r04/bench_default.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04u
mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
tail -2 $OUT/bench_default.err
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r04u/bench_default.json").read().strip().splitlines()[-1]); r = o["roofline"]
print(round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "fused", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 4), "issued", round(r["frac_issued"], 4),
      "other", round(r["other_ms_per_step"], 3), "pl", round(r["power_limited_mfma"]["frac_issued_of_power_limited"], 3), o["config"]["aggregation"])
for k, w in o["workloads"].items():
    print("  ", k, round(w["value"], 1), w.get("parity_linf"))
PY
