#!/bin/bash
# Round 4, GPU call H: bf16x3 vs fp16x3 (same MFMA rate, narrower multipliers: a power / clock data point), two streams, 64 graphs per GPU.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04h
mkdir -p $OUT
AB="--no-workloads --cpu-steps 0 --no-exact-fp32 --steps 20 --warmup 3"
for rep in 1 2; do
  timeout 300 python bench.py $AB > $OUT/ab_fp16x3_$rep.json 2> $OUT/ab_fp16x3_$rep.err
  timeout 300 python bench.py $AB --precision bf16x3 > $OUT/ab_bf16x3_$rep.json 2> $OUT/ab_bf16x3_$rep.err
  timeout 300 python bench.py $AB --streams 2 > $OUT/ab_streams2_$rep.json 2> $OUT/ab_streams2_$rep.err
done
timeout 600 python bench.py $AB --graphs-per-gpu 64 --steps 5 --warmup 2 > $OUT/bench_64graphs.json 2> $OUT/bench_64graphs.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04h/*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        r = o.get("roofline", {})
        print(f.split("/")[-1], round(o["value"], 1), "gs/s", round(o["ms_per_step"], 3), "ms/step  fused", round(r.get("avg_launch_ms", 0), 4),
              "other", round(r.get("other_ms_per_step", 0), 3), "repeats", [round(v, 3) for v in o.get("repeats", {}).get("ms_per_step", [])])
    except Exception as e:
        print(f, "unreadable", e)
PY
