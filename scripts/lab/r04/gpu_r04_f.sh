#!/bin/bash
# Round 4, GPU call F: synthetic-epilogue overlap experiment in the stage lab (serial at two waves per SIMD vs interleaved at one),
# the failing test of call E again, the default bench line with the dense workload.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04f
mkdir -p $OUT
timeout 600 python scripts/bench_stage_lab.py 142020,1142020,3142020,2142020,2143120,142020n,1142020n,2142020n,2143120n > $OUT/stage_lab_vmix.txt 2>&1; echo "lab exit $?" >> $OUT/stage_lab_vmix.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "finite or torch_custom or capturable" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_subset.log
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_decode.py -q > $OUT/pytest_r4_decode.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_r4_decode.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?" >> $OUT/bench_default.err
timeout 300 python bench.py --workload tsp50dense --steps 20 --warmup 5 --no-fusion --cpu-steps 0 --no-exact-fp32 > $OUT/bench_dense_unfused.json 2> $OUT/bench_dense_unfused.err
timeout 300 python bench.py --workload tsp50dense --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 > $OUT/bench_dense_fused.json 2> $OUT/bench_dense_fused.err
tail -12 $OUT/stage_lab_vmix.txt | cut -c1-200
tail -3 $OUT/pytest_subset.log; tail -3 $OUT/pytest_r4_decode.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04f/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])])
        for k, w in o.get("workloads", {}).items():
            print("   ", k, round(w["value"], 1), round(w["ms_per_step"], 3), [round(v, 3) for v in w["repeats"]["ms_per_step"]], w.get("parity_linf"))
    except Exception as e:
        print(f, "unreadable", e)
PY
