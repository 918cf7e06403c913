#!/bin/bash
# Round 5, GPU call M: the FINAL tree once more - whole GPU suite, smoke, end-to-end and decode benches, the driver's bench command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/test_gpu_all.txt 2>&1; echo "gpu suite exit $?"; tail -3 $OUT/test_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.txt
timeout 600 python scripts/bench_end_to_end.py > $OUT/bench_end_to_end.json 2> $OUT/bench_end_to_end.err; echo "e2e exit $?"; tail -c 900 $OUT/bench_end_to_end.json; echo
timeout 600 python scripts/bench_decode.py > $OUT/bench_decode.json 2> $OUT/bench_decode.err; echo "decode exit $?"; tail -c 600 $OUT/bench_decode.json; echo
BENCH_FULL_JSON=$OUT/bench_driver_cmd_full.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench exit $?"
python - <<'PY'
import json
lines = open("gpurun_out/r05m/bench_driver_cmd.json").read().strip().splitlines()
o = json.loads(lines[-1])
print("stdout lines", len(lines), "bytes", len(lines[-1]), "value", round(o["value"], 1), "rep", o["repeats"]["ms_per_step"], "fused", o["roofline"]["avg_launch_ms"],
      "other", o["roofline"]["other_ms_per_step"], "fabric", o["roofline"].get("fabric_frac_of_8TBs"), "power", o.get("power", {}).get("power_W_median"), "cpu", o["cpu_baseline"]["value"], "parity", o["parity_linf"])
for k, w in o["workloads"].items():
    print("  ", k, w["value"], w["rep_ms"], w["parity_linf"], w["other_ms"])
PY
