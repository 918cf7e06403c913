#!/bin/bash
# r06 call A: limiter flags (production step / zero weights / lab GEMM) + the driver's bench command with power.throttle
mkdir -p gpurun_out/r06
python scripts/lab/r06/limiter.py step 6 > gpurun_out/r06/limiter_step.txt 2> gpurun_out/r06/limiter_step.err
python scripts/lab/r06/limiter.py lab 6 > gpurun_out/r06/limiter_lab.txt 2> gpurun_out/r06/limiter_lab.err
BENCH_FULL_JSON=gpurun_out/r06/bench_call_a_full.json python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_call_a.json 2> gpurun_out/r06/bench_call_a.err
python3 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r06/bench_gpus2_refused.out 2>&1; echo "rc=$?" >> gpurun_out/r06/bench_gpus2_refused.out
grep -v "^{" gpurun_out/r06/limiter_step.txt; grep -v "^{" gpurun_out/r06/limiter_lab.txt; tail -c 1500 gpurun_out/r06/bench_call_a.json; tail -3 gpurun_out/r06/bench_gpus2_refused.out; tail -3 gpurun_out/r06/limiter_step.err
