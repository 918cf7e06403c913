#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/0,0/2,0/10" nostamp > $O/fused_ab.log 2>&1; cat $O/fused_ab.log
STAMP_VARIANT=21 timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/0" > $O/stamps_21.log 2>&1; grep -A9 "phase stamps" $O/stamps_21.log
for V in "--fused-opt 2" "--fused-opt 10"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-exact-fp32 $V 2>>$O/bench_ab.err | tee -a $O/bench_ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step fused', round(d['roofline']['avg_launch_ms'],4), 'ms')"
done
tail -3 $O/bench_ab.err
