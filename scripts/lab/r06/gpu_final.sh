#!/bin/bash
# r06 final-tree verification: the whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r06_final; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/test_gpu_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/test_gpu_all.txt; tail -3 $OUT/test_gpu_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
BENCH_FULL_JSON=$OUT/bench_driver_cmd_full.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r06_final/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("driver cmd:", o["value"], o["ms_per_step"], o["roofline"]["avg_launch_ms"], o["roofline"]["other_ms_per_step"], o["power"]["throttle"], {k: (v["value"], v["other_ms"]) for k, v in o["workloads"].items()}, len(open("gpurun_out/r06_final/bench_driver_cmd.json").read()))
PY
