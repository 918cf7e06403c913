#!/bin/bash
# Round 4, GPU call N (final tree): the whole GPU suite, smoke, the default bench line, and rocprofv3 --kernel-trace --stats of the
# default workload with enough launches for its average to be comparable with the bench's live HIP-event average.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04n
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
PB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --repeats 1"
timeout 300 python bench.py $PB > $OUT/bench_profiled_cmd_plain.json 2> $OUT/bench_profiled_cmd_plain.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py $PB > $REPO/$OUT/prof_stats.log 2>&1
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/r04n/prof_stats/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
with open("gpurun_out/r04n/rocprofv3_summary.txt", "w") as o:
    o.write(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s} {'min_us':>9s} {'max_us':>9s}\n")
    for r in rows[:18]:
        n = r["Name"].replace("void difusco::", "").replace("difusco::", "").split("(")[0][:100]
        o.write(f"{n:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} "
                f"{float(r['Percentage']):6.2f} {float(r['MinNs']) / 1e3:9.1f} {float(r['MaxNs']) / 1e3:9.1f}\n")
print(open("gpurun_out/r04n/rocprofv3_summary.txt").read()[:2500])
for f in ("bench_default", "bench_profiled_cmd_plain"):
    try:
        o = json.loads(open(f"gpurun_out/r04n/{f}.json").read().strip().splitlines()[-1]); r = o["roofline"]
        print(f, round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "fused avg ms", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 4),
              "issued", round(r["frac_issued"], 4), "other", round(r["other_ms_per_step"], 3), "pl", r.get("power_limited_mfma", {}).get("frac_issued_of_power_limited"))
        for k, w in (o.get("workloads") or {}).items():
            print("   ", k, round(w.get("value", 0), 1))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -2 $OUT/bench_default.err
