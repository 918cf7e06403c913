#!/bin/bash
# round 2, final evidence run: default bench, rocprofv3 stats + PMC for the headline run and for the other workloads' fused kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
bash scripts/gpu_profile.sh r02 tsp1000:800000:fused-fp16x3 > $O/profile_tsp1000.log 2>&1; tail -42 $O/profile_tsp1000.log | head -30
PROF_BENCH_ARGS="--workload mis" bash scripts/gpu_profile.sh r02mis mis:1328367:fused-fp16x3 > $O/profile_mis.log 2>&1; grep -A4 "pmc FETCH_SIZE\|pmc WRITE_SIZE" $O/profile_mis.log | head -14
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | cut -c1-300
timeout 600 python bench.py --workload mis --steps 10 --warmup 2 --cpu-steps 0 > $O/bench_mis.json 2>> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_mis.json')); print('mis', d['value'], d['roofline']['traffic'])"
