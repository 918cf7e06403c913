#!/bin/bash
# Round 4, GPU call E: re-read probe, clock / matrix-pipe counters of the GEMM-only lab kernel, PMC passes of the other workloads,
# issue-priority A/B, full GPU suite, smoke, the default bench line.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04e
mkdir -p $OUT
timeout 300 python scripts/bench_reread_probe.py > $OUT/reread_probe.txt 2>&1; echo "probe exit $?" >> $OUT/reread_probe.txt
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 python bench.py $AB --prof-lib > $OUT/ab_prio3_$rep.json 2> $OUT/ab_prio3_$rep.err
  timeout 300 python bench.py $AB --fused-opt 150899 > $OUT/ab_prio0_$rep.json 2> $OUT/ab_prio0_$rep.err
done
cd /tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $REPO/$OUT/pmc_lab -o lab -- python $REPO/scripts/bench_stage_lab.py 142020n,182020n,242020n > $REPO/$OUT/pmc_lab.log 2>&1
PB="--steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1"
for wl in tsp500 tsp10000 mis; do
  mkdir -p $REPO/$OUT/$wl
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/$OUT/$wl/pmc_$i -o bench -- python $REPO/bench.py $PB --workload $wl > $REPO/$OUT/$wl/pmc_$i.log 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/$wl/prof_stats -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1 --workload $wl > $REPO/$OUT/$wl/prof_stats.log 2>&1
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +8M -delete
for wl in tsp500 tsp10000 mis; do python scripts/summarize_pmc.py $OUT/$wl > $OUT/$wl/pmc_summary.txt 2>&1; done
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04e/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $OUT/reread_probe.txt | head -30
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
