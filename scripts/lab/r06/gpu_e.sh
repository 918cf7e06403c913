#!/bin/bash
# r06 call E: the table embedding kernel - tests, TSP-10000 with in-range (renoised) and free-running x_t, rocprofv3 kernel stats of both
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; OUT=gpurun_out/r06e; mkdir -p $OUT
python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -4
for mode in renoised free; do
  BENCH_FULL_JSON=$OUT/bench_tsp10000_${mode}_full.json python bench.py --workload tsp10000 --steps 10 --warmup 3 --cpu-steps 1 --no-exact-fp32 --gaussian-xt $mode 2>$OUT/bench_tsp10000_$mode.err | tail -1 > $OUT/bench_tsp10000_$mode.json
  python -c "
import json; d=json.load(open('$OUT/bench_tsp10000_${mode}_full.json')); print('$mode', d['value'], d['ms_per_step'], 'other', d['roofline']['other_ms_per_step'], 'parity', d.get('parity_linf'), 'max|xt|', d['config'].get('xt_abs_max_after_run'))"
done
cd /tmp
for mode in renoised free; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_$mode -o bench -- python $REPO/bench.py --workload tsp10000 --steps 5 --warmup 2 --cpu-steps 0 --no-exact-fp32 --repeats 1 --no-power --gaussian-xt $mode > $REPO/$OUT/prof_$mode.log 2> $REPO/$OUT/prof_$mode.err
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv, glob
for mode in ("renoised", "free"):
    for f in glob.glob(f"gpurun_out/r06e/prof_{mode}/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(f"gpurun_out/r06e/rocprofv3_summary_tsp10000_{mode}.txt", "w") as out:
            for r in rows[:12]:
                line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
                print(mode, line); out.write(line + "\n")
PY
