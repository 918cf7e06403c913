#!/bin/bash
# Round 5, GPU call B: the log2(e)-domain fused kernel (ABI 11).  New parity tests first, then a same-box A/B against the round-4
# tree (scratch/r04_tree: `git archive 49a5b37` + its own built library), interleaved; then the whole GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -s -m gpu > $OUT/test_gpu_round5.txt 2>&1; echo "round5 tests exit $?"
tail -25 $OUT/test_gpu_round5.txt
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads"
for rnd in 1 2 3; do
  (cd scratch/r04_tree && timeout 300 python bench.py $AB 2>/dev/null | tail -1 > ../../$OUT/ab_r04_tsp1000_$rnd.json)
  BENCH_FULL_JSON=$OUT/ab_r05_tsp1000_full_$rnd.json timeout 300 python bench.py $AB 2>/dev/null | tail -1 > $OUT/ab_r05_tsp1000_$rnd.json
done
for wl in tsp500 mis tsp10000; do
  for rnd in 1 2; do
    (cd scratch/r04_tree && timeout 300 python bench.py $AB --workload $wl 2>/dev/null | tail -1 > ../../$OUT/ab_r04_${wl}_$rnd.json)
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl 2>/dev/null | tail -1 > $OUT/ab_r05_${wl}_$rnd.json
  done
done
python - <<'PY'
import json, glob
for wl in ("tsp1000", "tsp500", "mis", "tsp10000"):
    for tag in ("r04", "r05"):
        vals = []
        for f in sorted(glob.glob(f"gpurun_out/r05b/ab_{tag}_{wl}_[0-9].json")):
            try:
                o = json.loads(open(f).read().strip().splitlines()[-1])
                vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4)))
            except Exception as e:
                vals.append(("ERR", str(e)[:40]))
        print(wl, tag, vals)
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/test_gpu_all.txt 2>&1; echo "gpu suite exit $?"
tail -8 $OUT/test_gpu_all.txt
