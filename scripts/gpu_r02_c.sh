#!/bin/bash
# round 2, GPU call C: software-pipelined stage body (OPT 6) A/B + phase stamps + bench A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/0,0/2,0/6,15/0,17/0,19/0" nostamp > $O/fused_ab.log 2>&1; cat $O/fused_ab.log
for V in 16 18 20; do
  STAMP_VARIANT=$V timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/0" > $O/stamps_$V.log 2>&1; echo "== stamps variant $V"; grep -A12 "phase stamps" $O/stamps_$V.log
done
for V in "--fused-opt 2" "--fused-opt 6" "--fused-opt 2" "--fused-opt 6"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-exact-fp32 $V 2>>$O/bench_ab.err | tee -a $O/bench_ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step fused', round(d['roofline']['avg_launch_ms'],4), 'ms')"
done
timeout 300 python -m pytest tests -m gpu -q -x -k "edge_layer_fused or golden_h256 or oracle_tsp_full_width" 2>&1 | tail -3
