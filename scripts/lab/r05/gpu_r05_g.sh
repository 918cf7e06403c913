#!/bin/bash
# Round 5, GPU call G: OPT bit 1 (alternating MFMA accumulator chains) on / off on the final kernel, every workload, interleaved
# (profiling library: the same translation units, variants selected by difusco_debug_set(7, ..)).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for wl in tsp1000 tsp500 mis tsp10000; do
  for rnd in 1 2 3; do
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl --prof-lib 2>/dev/null | tail -1 > $OUT/ab_${wl}_prod_$rnd.json
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl --fused-opt 151409 2>/dev/null | tail -1 > $OUT/ab_${wl}_151409_$rnd.json
  done
done
python - <<'PY'
import json, glob
for wl in ("tsp1000", "tsp500", "mis", "tsp10000"):
    for v in ("prod", "151409"):
        vals = []
        for f in sorted(glob.glob(f"gpurun_out/r05g/ab_{wl}_{v}_[0-9].json")):
            try:
                o = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4)))
            except Exception as e:
                vals.append(("ERR", str(e)[:60]))
        print(wl, v, vals)
PY
