#!/bin/bash
# Round 5, GPU call L: sincosf per feature pair in the generated-embedding GEMM (Gaussian / non-binary inputs): parity tests that reach it,
# TSP-10000 before / after is read from `other_ms_per_step` (call H: 1.66 ms), then the default line.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05l
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "gauss or Gauss or binary or golden or h256 or oracle or adversarial or class" > $OUT/tests_subset.txt 2>&1; echo "tests exit $?"; tail -3 $OUT/tests_subset.txt
for rnd in 1 2; do
  BENCH_FULL_JSON=/dev/null timeout 600 python bench.py --workload tsp10000 --steps 10 --warmup 3 --cpu-steps 1 --no-exact-fp32 --no-power 2>/dev/null | tail -1 > $OUT/bench_tsp10000_$rnd.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05l/bench_tsp10000_*.json")):
    o = json.loads(open(f).read().strip().splitlines()[-1]); r = o["roofline"]
    print(f.split("/")[-1], round(o["value"], 2), "ms/step", round(o["ms_per_step"], 3), "fused", r["avg_launch_ms"], "other", r["other_ms_per_step"], "parity", o.get("parity_linf"))
PY
