#!/bin/bash
# production = iterative-maxocc scheduler + OPT 371: full GPU suite, default bench first (fresh box), then profiles, workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/o5; mkdir -p $O
timeout 600 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json; cut -c1-400 $O/bench_default.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for wl in tsp500 tsp10000 mis; do
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-steps 0 --no-exact-fp32 2>/dev/null | grep '^{' > $O/bench_workload_$wl.json
  python -c "import json; d=json.load(open('$O/bench_workload_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
timeout 600 python bench.py --graphs-per-gpu 64 --steps 10 --warmup 2 --cpu-steps 0 --no-exact-fp32 2>/dev/null | grep '^{' > $O/bench_tsp1000_64graphs.json
python -c "import json; d=json.load(open('$O/bench_tsp1000_64graphs.json')); print('x64', d['value'], d['ms_per_step'])"
bash scripts/gpu_profile.sh r02 tsp1000:800000:fused-fp16x3 > $O/profile_tsp1000.log 2>&1; tail -48 $O/profile_tsp1000.log | head -40
PROF_BENCH_ARGS="--workload mis" bash scripts/gpu_profile.sh r02mis mis:1328367:fused-fp16x3 > $O/profile_mis.log 2>&1; grep -A4 "pmc FETCH_SIZE\|pmc WRITE_SIZE" $O/profile_mis.log | head -14
