#!/bin/bash
# Round 4, GPU call A: stage-loop lab, the new parity tests, the default bench line (+ binding / prepared-state A/B), rocprofv3
# stats of the default run.   usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r04_a.sh'
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 600 python scripts/bench_stage_lab.py > $OUT/stage_lab.txt 2>&1; echo "lab exit $?" >> $OUT/stage_lab.txt
timeout 1500 python -m pytest tests/test_gpu_round4.py -q -s --maxfail=20 > $OUT/pytest_round4.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_round4.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 python bench.py $AB > $OUT/ab_ctypes_$rep.json 2> $OUT/ab_ctypes_$rep.err
  timeout 300 python bench.py $AB --backend torch > $OUT/ab_torch_$rep.json 2> $OUT/ab_torch_$rep.err
  timeout 300 python bench.py $AB --no-prepare > $OUT/ab_noprepare_$rep.json 2> $OUT/ab_noprepare_$rep.err
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1 > $REPO/$OUT/prof_stats.log 2>&1
echo "stats exit $?" >> $REPO/$OUT/prof_stats.log
cd $REPO
find $OUT/prof_stats -name "*kernel_trace.csv" -size +20M -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04a/ab_*.json")) + ["gpurun_out/r04a/bench_default.json"]:
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "host us/step", round(o.get("host_enqueue_us_per_step", 0), 1),
              "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -25 $OUT/stage_lab.txt | cut -c1-200
grep -E "L_inf|passed|failed|FAILED|Error|philox|DDPM|N = " $OUT/pytest_round4.log | tail -30
