#!/bin/bash
# Round 4, GPU call L: OPT bit 18 of the fused kernel (planes of a product straight from its factors, v_fma_mix{lo,hi}_f16):
# parity subset on the new default library, A/B against the build without the bit (variant library opt_151411), and a
# socket-power / clock trace (rocm-smi) next to a long run of the default bench - direct evidence for the power-limit claim.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04l
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q --maxfail=10 -k "golden or oracle or dense or bench_workload or prepared or fixture" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_subset.log
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3 --backend ctypes"
for rep in 1 2 3; do
  timeout 300 python bench.py $AB > $OUT/ab_bit18_$rep.json 2> $OUT/ab_bit18_$rep.err
  DIFUSCO_HIP_LIBRARY=$PWD/difusco_amd/lib/libdifusco_hip_opt_151411.so timeout 300 python bench.py $AB > $OUT/ab_nobit18_$rep.json 2> $OUT/ab_nobit18_$rep.err
done
# the other workloads, once each way
for wl in tsp500 mis tsp10000; do
  timeout 300 python bench.py $AB --workload $wl > $OUT/ab_bit18_$wl.json 2> $OUT/ab_bit18_$wl.err
  DIFUSCO_HIP_LIBRARY=$PWD/difusco_amd/lib/libdifusco_hip_opt_151411.so timeout 300 python bench.py $AB --workload $wl > $OUT/ab_nobit18_$wl.json 2> $OUT/ab_nobit18_$wl.err
done
# power / clock trace: idle sample, then samples every 0.5 s while the default workload runs 150 steps x 3 repetitions
rocm-smi -M > $OUT/power_cap.txt 2>&1
rocm-smi -P -g > $OUT/power_idle.txt 2>&1
( timeout 300 python bench.py --no-workloads --cpu-steps 0 --no-exact-fp32 --steps 150 --warmup 5 > $OUT/bench_long.json 2> $OUT/bench_long.err ) &
BPID=$!
: > $OUT/power_trace.txt
for i in $(seq 1 60); do
  if ! kill -0 $BPID 2>/dev/null; then break; fi
  echo "--- sample $i $(date +%s.%N)" >> $OUT/power_trace.txt
  rocm-smi -P -g 2>&1 | grep -E "Power|sclk|GPU\[" >> $OUT/power_trace.txt
  sleep 0.5
done
wait $BPID
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04l/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1]); r = o["roofline"]
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "fused avg ms", round(r["avg_launch_ms"], 4), "other", round(r["other_ms_per_step"], 3))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -2 $OUT/pytest_subset.log
cat $OUT/power_cap.txt | grep -i -E "power|W" | head -3
grep -E "Power" $OUT/power_trace.txt | head -40 | tr '\n' ';' | cut -c1-1500
