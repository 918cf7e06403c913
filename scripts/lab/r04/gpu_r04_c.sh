#!/bin/bash
# Round 4, GPU call C: packed-fp32 A/B (stage lab + whole step, three builds of the library), 1-workgroup-per-CU phase stamps,
# the round-4 tests again.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04c
mkdir -p $OUT
timeout 600 python scripts/bench_stage_lab.py 142020,142020n,182020,182020n,242020,242020n,143120,143120n > $OUT/stage_lab_pk.txt 2>&1; echo "lab exit $?" >> $OUT/stage_lab_pk.txt
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 python bench.py $AB > $OUT/ab_prod_$rep.json 2> $OUT/ab_prod_$rep.err
  DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_nopk_fused.so timeout 300 python bench.py $AB > $OUT/ab_nopkfused_$rep.json 2> $OUT/ab_nopkfused_$rep.err
  DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_nopk_all.so timeout 300 python bench.py $AB > $OUT/ab_nopkall_$rep.json 2> $OUT/ab_nopkall_$rep.err
done
for wl in tsp500 mis tsp10000; do
  timeout 300 python bench.py $AB --workload $wl --steps 10 > $OUT/wl_${wl}_prod.json 2> $OUT/wl_${wl}_prod.err
  DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_nopk_all.so timeout 300 python bench.py $AB --workload $wl --steps 10 > $OUT/wl_${wl}_nopkall.json 2> $OUT/wl_${wl}_nopkall.err
done
DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_nopk_all.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -s --maxfail=20 -k "oracle or golden or bench_workload or posterior or fused" > $OUT/pytest_nopk.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_nopk.log
DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_nopk_all.so timeout 900 python -m pytest tests/test_gpu_round4.py -q -s --maxfail=20 > $OUT/pytest_nopk_r4.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_nopk_r4.log
timeout 900 python -m pytest tests/test_gpu_round4.py -q -s --maxfail=20 > $OUT/pytest_round4.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_round4.log
LDS_PAD=4000 STAMP_VARIANT=18 timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/20339" > $OUT/stamps_1wg.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04c/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])], o["config"].get("binding"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -12 $OUT/stage_lab_pk.txt | cut -c1-200
grep -E "passed|failed|FAILED|Error" $OUT/pytest_nopk.log | tail; grep -E "passed|failed|FAILED|Error" $OUT/pytest_nopk_r4.log | tail; grep -E "passed|failed|FAILED|Error" $OUT/pytest_round4.log | tail
grep -E "median|phase stamps|ticks" $OUT/stamps_1wg.txt | head -14
