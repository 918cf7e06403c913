#!/bin/bash
# Round 5, GPU call K: rocprofv3 stats + PMC passes of the default run on the FINAL production set (151409), and the default bench line of that box.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
PB="--steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1 --no-power"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --repeats 1 --no-power > $REPO/$OUT/prof_stats.log 2> $REPO/$OUT/prof_stats.err
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -o bench -- python $REPO/bench.py $PB > $REPO/$OUT/pmc_$i.log 2>&1
  echo "pmc set $i exit $?" >> $REPO/$OUT/pmc_sets.txt
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r05k/prof_stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("gpurun_out/r05k/rocprofv3_kernel_stats.txt", "w") as out:
        for r in rows[:12]:
            line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
            print(line); out.write(line + "\n")
PY
tail -1 $OUT/prof_stats.log | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('profiled run: live avg_launch_ms', o['roofline']['avg_launch_ms'], 'value', o['value'])"
BENCH_FULL_JSON=$OUT/bench_full.json timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
python -c "
import json; o=json.loads(open('gpurun_out/r05k/bench_default.json').read().strip().splitlines()[-1]); print(o['value'], o['roofline']['avg_launch_ms'], o['repeats']['ms_per_step'], o['power'])"
cat $OUT/pmc_sets.txt
