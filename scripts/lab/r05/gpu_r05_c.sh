#!/bin/bash
# Round 5, GPU call C: evidence for the log2(e)-domain kernel - lo-plane power lab, rocprofv3 stats + PMC passes of the default run
# (SQ instruction counts before/after, traffic), the round-4 bench.py on the dense sub-workload alone (its first-repetition
# transient), the default bench line.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
timeout 200 python scripts/bench_lab_lo_bits.py > $OUT/lab_lo_bits.txt 2>&1; echo "lab exit $?"
cat $OUT/lab_lo_bits.txt | grep -v "^{" | tail -14
# the round-4 tree, dense sub-workload alone and inside its default run order (mis, then tsp50dense)
(cd scratch/r04_tree && timeout 200 python bench.py --workload tsp50dense --cpu-steps 0 --no-exact-fp32 --steps 10 --warmup 4 2>/dev/null | tail -1 > ../../$OUT/r04_bench_tsp50dense_alone.json)
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r05c/r04_bench_tsp50dense_alone.json").read().strip().splitlines()[-1])
print("round-4 bench.py, tsp50dense alone: repetitions ms/step", [round(v, 3) for v in o["repeats"]["ms_per_step"]])
PY
# ---- profiles of the default run
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
  echo "pmc set $i ($SET) exit $?" >> $REPO/$OUT/pmc_sets.txt
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r05c/prof_stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
with open("gpurun_out/r05c/rocprofv3_kernel_stats.txt", "w") as out:
    for r in rows[:16]:
        line = f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs']) / 1e6:10.3f} avg_us {float(r['AverageNs']) / 1e3:9.2f} pct {r['Percentage']}"
        print(line); out.write(line + "\n")
PY
tail -1 $OUT/prof_stats.log | head -c 1500; echo
BENCH_FULL_JSON=$OUT/bench_full.json timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
tail -c 3900 $OUT/bench_default.json
cat $OUT/pmc_sets.txt
grep -A12 "edge_layer_fused_kernel<FFp16, 0, 4, false, false, 0" $OUT/pmc_summary.txt | head -60
