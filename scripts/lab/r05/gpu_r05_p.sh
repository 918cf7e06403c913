#!/bin/bash
# Round 5, GPU call P: rocprofv3 kernel stats of the TSP-10000 Gaussian workload (the new embedding kernel's duration).
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r05p
mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --workload tsp10000 --steps 5 --warmup 2 --cpu-steps 0 --no-exact-fp32 --repeats 1 --no-power > $REPO/$OUT/prof_stats.log 2> $REPO/$OUT/prof_stats.err
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r05p/prof_stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("gpurun_out/r05p/rocprofv3_kernel_stats_tsp10000.txt", "w") as out:
        for r in rows[:12]:
            line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
            print(line); out.write(line + "\n")
PY
