#!/bin/bash
# Round 4, GPU call D: A/B of the neighbour-sum fast path and of the packed-fp32 build on the final library, PMC passes (SQ, traffic,
# L2 / EA level counters) and rocprofv3 stats of the default run, round-4 tests, default bench line.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04d
mkdir -p $OUT
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 python bench.py $AB --prof-lib > $OUT/ab_aggfast_$rep.json 2> $OUT/ab_aggfast_$rep.err
  timeout 300 python bench.py $AB --fused-opt 20339 > $OUT/ab_noaggfast_$rep.json 2> $OUT/ab_noaggfast_$rep.err
  timeout 300 python bench.py $AB > $OUT/ab_prod_$rep.json 2> $OUT/ab_prod_$rep.err
  DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_pk_fused.so timeout 300 python bench.py $AB > $OUT/ab_pkfused_$rep.json 2> $OUT/ab_pkfused_$rep.err
done
for wl in tsp500 mis; do
  timeout 300 python bench.py $AB --workload $wl --steps 10 --prof-lib > $OUT/wl_${wl}_aggfast.json 2> $OUT/wl_${wl}_aggfast.err
  timeout 300 python bench.py $AB --workload $wl --steps 10 --fused-opt 20339 > $OUT/wl_${wl}_noaggfast.json 2> $OUT/wl_${wl}_noaggfast.err
done
timeout 900 python -m pytest tests/test_gpu_round4.py -q -s --maxfail=20 > $OUT/pytest_round4.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_round4.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q --maxfail=20 -k "fused or oracle or golden or bench_workload or posterior or trajectory" > $OUT/pytest_parity_subset.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_parity_subset.log
# ---- profiles of the default run
PB="--steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_stats -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-steps 0 --no-profile --no-exact-fp32 --no-workloads --repeats 1 > $REPO/$OUT/prof_stats.log 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$i -o bench -- python $REPO/bench.py $PB > $REPO/$OUT/pmc_$i.log 2>&1
  echo "pmc set $i ($SET) exit $?" >> $REPO/$OUT/pmc_sets.txt
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python scripts/summarize_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04d/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])], o["config"].get("binding"))
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -E "passed|failed|FAILED|Error" $OUT/pytest_round4.log | tail -5; grep -E "passed|failed|FAILED|Error" $OUT/pytest_parity_subset.log | tail -5
cat $OUT/pmc_sets.txt; head -80 $OUT/pmc_summary.txt
