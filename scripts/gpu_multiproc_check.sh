#!/bin/bash
# Exercise bench.py's multi-process path (rank sharding, weight broadcast, barriers, max-over-ranks timing) on a
# ONE-GPU box: 2 ranks share GPU 0 over gloo.  RCCL itself is exercised by the driver's multi-GPU runs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --graphs-per-gpu 4 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
echo "exit $?"; cat gpurun_out/bench_2rank_gloo.json | cut -c1-400; tail -5 gpurun_out/bench_2rank_gloo.err
