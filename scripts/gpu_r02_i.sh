#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/19,0/51,0/83,0/115" nostamp 2>&1 | grep -v amdgpu.ids | tee $O/fused_ab.log
STAMP_VARIANT=26 timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/19" > $O/stamps_26.log 2>&1; grep -A9 "phase stamps" $O/stamps_26.log
for V in "--fused-opt 19" "--fused-opt 115" "--fused-opt 19" "--fused-opt 115"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-exact-fp32 $V 2>>$O/bench_ab.err | tee -a $O/bench_ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step fused', round(d['roofline']['avg_launch_ms'],4), 'ms')"
done
timeout 600 python -m pytest tests -m gpu -q -x -k "knn or edge_layer_fused or golden_h256 or bench_workload" 2>&1 | tail -3
