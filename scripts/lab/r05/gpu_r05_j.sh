#!/bin/bash
# Round 5, GPU call J: code-generation A/Bs on the final kernel - no scheduling fences around the MFMA triples (OPT bit 21, profiling
# library) and the fused translation units under LLVM's other scheduling strategies (variant builds of the production sources).
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for rnd in 1 2; do
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --prof-lib 2>/dev/null | tail -1 > $OUT/ab_proflib_$rnd.json
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --fused-opt 2248561 2>/dev/null | tail -1 > $OUT/ab_freesched_$rnd.json
  BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB --backend ctypes 2>/dev/null | tail -1 > $OUT/ab_production_$rnd.json
  for v in sched_maxilp sched_memclause sched_iterilp; do
    DIFUSCO_HIP_LIBRARY=$REPO/difusco_amd/lib/libdifusco_hip_$v.so BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB 2>/dev/null | tail -1 > $OUT/ab_${v}_$rnd.json
  done
done
python - <<'PY'
import json, glob
for v in ("proflib", "freesched", "production", "sched_maxilp", "sched_memclause", "sched_iterilp"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r05j/ab_{v}_[0-9].json")):
        try:
            o = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4), o["config"]["binding"][:6]))
        except Exception as e:
            vals.append(("ERR", str(e)[:60]))
    print(v, vals)
PY
