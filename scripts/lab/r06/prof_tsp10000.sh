#!/bin/bash
# rocprofv3 kernel stats of the TSP-10000 Gaussian workload (x_t ~ N(0,1)): the table embedding kernel's duration.  usage: prof_tsp10000.sh <tag>
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; OUT=gpurun_out/r06_$1; mkdir -p $OUT
python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o bench -- python $REPO/bench.py --workload tsp10000 --steps 5 --warmup 2 --cpu-steps 0 --no-exact-fp32 --repeats 1 --no-power > $REPO/$OUT/prof.log 2>&1
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - $OUT <<'PY'
import csv, glob, sys
out_dir = sys.argv[1]
for f in glob.glob(out_dir + "/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out_dir + "/rocprofv3_summary_tsp10000.txt", "w") as out:
        for r in rows[:12]:
            line = "%-100s calls %6s total_ms %10.3f avg_us %9.2f pct %s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"])
            print(line[:190]); out.write(line + "\n")
PY
