#!/bin/bash
# final default bench line of round 2 (fresh box) + phase stamps of the production kernel under the new scheduler
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/q; mkdir -p $O
timeout 600 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json; cut -c1-200 $O/bench_default.json
timeout 600 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-exact-fp32 2>/dev/null | grep '^{' > $O/bench_30.json
python -c "import json; r=json.load(open('$O/bench_30.json')); print('30 steps', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['other_ms_per_step'])"
timeout 600 python bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-exact-fp32 --fused-opt 0 --no-node-reorder --node-linear-depth 1 2>/dev/null | grep '^{' > $O/bench_r1like.json
python -c "import json; r=json.load(open('$O/bench_r1like.json')); print('r1-like', r['value'], r['ms_per_step'])"
for V in 18 24; do
  STAMP_VARIANT=$V timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/371" > $O/stamps_$V.log 2>&1; echo "== stamps variant $V"; grep -A9 "phase stamps" $O/stamps_$V.log
done
LDS_PAD=70000 timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/371" > $O/one_wg_per_cu.log 2>&1; tail -4 $O/one_wg_per_cu.log
