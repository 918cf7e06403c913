#!/bin/bash
# Round 5, GPU call F: option A/Bs on the final kernel (profiling library: production set, without the alternating MFMA chains, without the
# two-stage cover of the e stream), and bench.py --gpus 2 with two ranks on the one GPU of the box over gloo (plumbing of the N > 1 line).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
AB="--steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads --no-power"
for rnd in 1 2; do
  for v in prod 151409 151379; do
    if [ $v = prod ]; then FL="--prof-lib"; else FL="--fused-opt $v"; fi
    BENCH_FULL_JSON=/dev/null timeout 300 python bench.py $AB $FL 2>/dev/null | tail -1 > $OUT/ab_${v}_$rnd.json
  done
done
for gn in per_shard_call global; do
  BENCH_FULL_JSON=/dev/null BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 10 --warmup 3 --graphs-per-gpu 4 --gn-stats $gn --cpu-steps 0 --no-exact-fp32 --no-workloads 2> $OUT/bench_2ranks_1gpu_gloo_$gn.err | tail -1 > $OUT/bench_2ranks_1gpu_gloo_$gn.json
done
python - <<'PY'
import json, glob
for v in ("prod", "151409", "151379"):
    vals = []
    for f in sorted(glob.glob(f"gpurun_out/r05f/ab_{v}_[0-9].json")):
        try:
            o = json.loads(open(f).read().strip().splitlines()[-1]); vals.append((round(o["value"], 1), round(o["roofline"]["avg_launch_ms"], 4)))
        except Exception as e:
            vals.append(("ERR", str(e)[:60]))
    print(v, vals)
for gn in ("per_shard_call", "global"):
    try:
        o = json.loads(open(f"gpurun_out/r05f/bench_2ranks_1gpu_gloo_{gn}.json").read().strip().splitlines()[-1])
        print(gn, "n_gpus", o["n_gpus"], "ranks_seen", o["ranks_seen"], "rank_ms", o["rank_ms_per_step"], "value", round(o["value"], 1), "broadcast_ms", o["broadcast_ms"], "bytes", len(json.dumps(o)))
    except Exception as e:
        print(gn, "ERR", e)
PY
