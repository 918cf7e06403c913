#!/bin/bash
# round 2, GPU call A: parity tests (new H=256 goldens, bench-workload tests), kernel variant A/B, bench A/B.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -s > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -15
timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/0,0/1,0/2,0/3,15/0,17/0" nostamp > $O/fused_ab_morton.log 2>&1
NODE_ORDER=caller timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/0,0/1,0/2,0/3" nostamp > $O/fused_ab_caller.log 2>&1
cat $O/fused_ab_morton.log $O/fused_ab_caller.log
for V in "--fused-opt 0 --no-node-reorder" "--fused-opt 0" "--fused-opt 1" "--fused-opt 2" "--fused-opt 3"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --no-exact-fp32 $V 2>>$O/bench_ab.err | tee -a $O/bench_ab.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step fused', round(d['roofline']['avg_launch_ms'],4), 'ms')"
done
timeout 600 python bench.py --steps 20 --warmup 3 --fused-opt 3 > $O/bench_full.json 2> $O/bench_full.err
cat $O/bench_full.json; tail -3 $O/bench_full.err
