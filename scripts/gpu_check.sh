#!/bin/bash
# GPU-side check: parity tests, smoke, one bench line.  Usage: gpurun -- 'bash scripts/gpu_check.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 600 python bench.py ${BENCH_ARGS:---steps 10 --warmup 2} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
grep -E "L_inf|passed|failed|FAILED|Error" gpurun_out/pytest_gpu.log | tail -40; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
