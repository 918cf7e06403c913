#!/bin/bash
# r06 call D: the whole GPU suite on the tree, the driver's bench command, the upper bound of any node-kernel fold,
# rocprofv3 kernel stats of the default run and of the TSP-10000 Gaussian workload (table embedding).
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; OUT=gpurun_out/r06d; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/test_gpu_all.txt 2>&1; echo "pytest rc=$?" >> $OUT/test_gpu_all.txt; tail -3 $OUT/test_gpu_all.txt
BENCH_FULL_JSON=$OUT/bench_driver_cmd_full.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
# upper bound of folding the node kernels: the step with their launches SKIPPED (wrong results, timing only), same box, interleaved
AB="--prof-lib --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for rnd in 1 2; do
  for v in "0" "1" "2" "3"; do
    BENCH_FULL_JSON=/dev/null python3 bench.py $AB --debug-set 11=$v 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('skip=$v round=$rnd value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'fused ms/launch', round(d['roofline']['avg_launch_ms'], 4), 'other', round(d['roofline']['other_ms_per_step'], 4), 'reps', d['repeats']['ms_per_step'])" >> $OUT/ab_node_kernels_skipped.txt
  done
done
cat $OUT/ab_node_kernels_skipped.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --repeats 1 --no-power > $REPO/$OUT/prof_stats.log 2> $REPO/$OUT/prof_stats.err
cd $REPO; python scripts/lab/r06/limiter.py lab 5 > $OUT/limiter_lab_fp8.txt 2> $OUT/limiter_lab_fp8.err; grep -v "^{" $OUT/limiter_lab_fp8.txt
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv, glob, json
for tag, name in (("prof_stats", "rocprofv3_summary_tsp1000_default.txt"),):
    for f in glob.glob(f"gpurun_out/r06d/{tag}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(f"gpurun_out/r06d/{name}", "w") as out:
            for r in rows[:14]:
                line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
                print(line); out.write(line + "\n")
for tag in ("prof_stats",):
    try:
        o = json.loads(open(f"gpurun_out/r06d/{tag}.log").read().strip().splitlines()[-1])
        print(tag, "profiled run: live avg_launch_ms", o["roofline"]["avg_launch_ms"], "value", o["value"])
    except Exception as exc:
        print(tag, exc)
o = json.loads(open("gpurun_out/r06d/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("driver cmd:", o["value"], o["ms_per_step"], o["roofline"]["avg_launch_ms"], o["roofline"]["other_ms_per_step"], o["power"], {k: (v["value"], v["other_ms"]) for k, v in o["workloads"].items()})
PY
