#!/bin/bash
# Round 5, GPU call D: same-box A/B of the 8-row-group boundary tests in the neighbour sum (scratch/r05a_tree = the tree before it),
# parity tests of the aggregation paths, and the issue-priority A/B on the new instruction mix.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_aggregation.py tests/test_gpu_round4.py -x -q -m gpu > $OUT/tests_subset.txt 2>&1; echo "tests exit $?"; tail -3 $OUT/tests_subset.txt
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for rnd in 1 2 3; do
  (cd scratch/r05a_tree && BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB 2>/dev/null | tail -1 > ../../$OUT/ab_before_tsp1000_$rnd.json)
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB 2>/dev/null | tail -1 > $OUT/ab_after_tsp1000_$rnd.json
done
for wl in tsp500 mis; do
  for rnd in 1 2; do
    (cd scratch/r05a_tree && BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl 2>/dev/null | tail -1 > ../../$OUT/ab_before_${wl}_$rnd.json)
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl 2>/dev/null | tail -1 > $OUT/ab_after_${wl}_$rnd.json
  done
done
for rnd in 1 2; do
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --prof-lib 2>/dev/null | tail -1 > $OUT/ab_proflib_prod_$rnd.json
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --fused-opt 150899 2>/dev/null | tail -1 > $OUT/ab_proflib_noprio_$rnd.json
done
python - <<'PY'
import json, glob
for pat in ("before_tsp1000", "after_tsp1000", "before_tsp500", "after_tsp500", "before_mis", "after_mis", "proflib_prod", "proflib_noprio"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r05d/ab_{pat}_[0-9].json")):
        try:
            o = json.loads(open(f).read().strip().splitlines()[-1])
            vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4)))
        except Exception as e:
            vals.append(("ERR", str(e)[:40]))
    print(pat, vals)
PY
