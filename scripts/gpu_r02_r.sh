#!/bin/bash
# last-layer GroupNorm sums through the scratch: parity, stamps of the last-layer variant, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider > $O/parity.log 2>&1; tail -1 $O/parity.log
for V in 32; do
  STAMP_VARIANT=$V timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/1907" > $O/stamps_$V.log 2>&1; echo "== stamps variant $V"; grep -A9 "phase stamps" $O/stamps_$V.log
done
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-exact-fp32 --cpu-steps 0 --profile-all 2>/dev/null | grep '^{' > $O/bench.json
python - <<PY
import json; r=json.load(open("$O/bench.json")); print(r["value"], r["ms_per_step"], r["roofline"]["frac"])
k=r.get("kernels") or r["roofline"].get("kernels")
print(str(k)[:600] if k else list(r.keys()))
PY
done
