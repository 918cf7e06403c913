#!/bin/bash
# Round 5, GPU call I: GEMM 2 block-major inside a stage (OPT bit 20: accumulator chains of 12 MFMAs instead of 3) vs the production order,
# profiling library, interleaved.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for wl in tsp1000 mis; do
  for rnd in 1 2 3; do
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl --prof-lib 2>/dev/null | tail -1 > $OUT/ab_${wl}_prod_$rnd.json
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --workload $wl --fused-opt 1199985 2>/dev/null | tail -1 > $OUT/ab_${wl}_1199985_$rnd.json
  done
done
python - <<'PY'
import json, glob
for wl in ("tsp1000", "mis"):
    for v in ("prod", "1199985"):
        vals = []
        for f in sorted(glob.glob(f"gpurun_out/r05i/ab_{wl}_{v}_[0-9].json")):
            try:
                o = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4), o["parity_linf"] if "parity_linf" in o else None))
            except Exception as e:
                vals.append(("ERR", str(e)[:60]))
        print(wl, v, vals)
PY
