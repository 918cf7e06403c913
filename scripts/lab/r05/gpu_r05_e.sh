#!/bin/bash
# Round 5, GPU call E: final tree - whole GPU suite, smoke, the driver's bench command, PMC passes (traffic + SQ) of the other
# workloads, rocprofv3 stats of the final default run.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/test_gpu_all.txt 2>&1; echo "gpu suite exit $?"; tail -3 $OUT/test_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.txt
BENCH_FULL_JSON=$OUT/bench_driver_cmd_full.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench exit $?"
python - <<'PY'
import json
s = open("gpurun_out/r05e/bench_driver_cmd.json").read()
lines = s.strip().splitlines()
o = json.loads(lines[-1])
print("stdout lines", len(lines), "bytes of the line", len(lines[-1]), "value", round(o["value"], 1), "rep", o["repeats"]["ms_per_step"], "fused", o["roofline"]["avg_launch_ms"],
      "other", o["roofline"]["other_ms_per_step"], "power", o.get("power", {}).get("power_W_median"), "cpu", o["cpu_baseline"]["value"])
for k, w in o["workloads"].items():
    print("  ", k, w["value"], w["rep_ms"], w["parity_linf"], w["other_ms"])
PY
PB="--steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1 --no-power"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --repeats 1 --no-power > $REPO/$OUT/prof_stats.log 2> $REPO/$OUT/prof_stats.err
for wl in tsp500 mis tsp10000; do
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"; do
    i=$((i+1))
    mkdir -p $REPO/$OUT/$wl
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/$OUT/$wl/pmc_$i -o bench -- python $REPO/bench.py $PB --workload $wl > $REPO/$OUT/$wl/pmc_$i.log 2>&1
    echo "$wl pmc set $i exit $?" >> $REPO/$OUT/pmc_sets.txt
  done
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
for wl in tsp500 mis tsp10000; do python scripts/summarize_pmc.py $OUT/$wl > $OUT/$wl/pmc_summary.txt 2>&1; done
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r05e/prof_stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("gpurun_out/r05e/rocprofv3_kernel_stats.txt", "w") as out:
        for r in rows[:12]:
            line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
            print(line); out.write(line + "\n")
PY
tail -1 $OUT/prof_stats.log | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('profiled run: live avg_launch_ms', o['roofline']['avg_launch_ms'], 'value', o['value'])"
cat $OUT/pmc_sets.txt
