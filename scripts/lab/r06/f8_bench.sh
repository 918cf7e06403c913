#!/bin/bash
# (branch fp8c only) fp16f8 vs fp16x3: accuracy check + interleaved bench lines with power / clock / PPT residency
cd $GRAFT_REPO_ROOT
python scripts/lab/r06/f8_mode_check.py 2>&1 | grep "fp16f8"
for rnd in 1 2; do for p in fp16x3 fp16f8; do BENCH_FULL_JSON=/dev/null python bench.py --precision $p --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 --no-workloads 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); pw=d['power']; print('$p', round(d['value'],1), 'ms/launch', d['roofline']['avg_launch_ms'], 'other', round(d['roofline']['other_ms_per_step'],3), 'W', pw['power_W_median'], 'MHz', pw['sclk_MHz_median'], 'ppt', pw['throttle']['ppt'], 'J/gs', round(pw['J_per_graph_step'],3))"; done; done
