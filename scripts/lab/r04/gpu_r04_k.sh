#!/bin/bash
# Round 4, GPU call K: head kernel with non-temporal reads of e (A/B against the previous build kept as a variant library), tests of the head.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04k
mkdir -p $OUT
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3 --profile-all"
for rep in 1 2; do
  timeout 300 python bench.py $AB > $OUT/ab_headnt_$rep.json 2> $OUT/ab_headnt_$rep.err
  DIFUSCO_HIP_LIBRARY=$PWD/difusco_amd/lib/libdifusco_hip_headdefault.so timeout 300 python bench.py $AB > $OUT/ab_headdefault_$rep.json 2> $OUT/ab_headdefault_$rep.err
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -q --maxfail=10 -k "golden or oracle or dense or bench_workload or prepared" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_subset.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04k/*.json")):
    o = json.loads(open(f).read().strip().splitlines()[-1]); r = o["roofline"]; k = o["kernels"]
    print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "other", round(r["other_ms_per_step"], 3), "head ms/step", round(k["head"]["ms_total"] / (o["steps"] * o["repeats"]["n"]), 4))
PY
tail -2 $OUT/pytest_subset.log
