#!/bin/bash
# Round 5, GPU call N: edge_embed_tiled_kernel (the generated-embedding GEMM of Gaussian / non-binary steps in the fused kernel's GEMM 1
# dataflow): parity tests that reach it, TSP-10000 line (other_ms_per_step: 1.66 before the sincosf change, 1.54 after it).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05n
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -m gpu -k "gauss or Gauss or binary or golden or h256 or oracle or adversarial or class" > $OUT/tests_subset.txt 2>&1; echo "tests exit $?"; tail -4 $OUT/tests_subset.txt
for rnd in 1 2; do
  BENCH_FULL_JSON=/dev/null timeout 600 python bench.py --workload tsp10000 --steps 10 --warmup 3 --cpu-steps 1 --no-exact-fp32 --no-power 2>/dev/null | tail -1 > $OUT/bench_tsp10000_$rnd.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05n/bench_tsp10000_*.json")):
    o = json.loads(open(f).read().strip().splitlines()[-1]); r = o["roofline"]
    print(f.split("/")[-1], round(o["value"], 2), "ms/step", round(o["ms_per_step"], 3), "fused", r["avg_launch_ms"], "other", r["other_ms_per_step"], "parity", o.get("parity_linf"))
PY
