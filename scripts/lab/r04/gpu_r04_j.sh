#!/bin/bash
# Round 4, GPU call J: bench.py --gpus 2 under torch.distributed.run with two ranks sharing the one GPU of the box (gloo; plumbing
# check of the N > 1 JSON fields with real kernels, not a scaling number), both statistics modes.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04j
mkdir -p $OUT
for gn in per_shard_call global; do
  BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 10 --warmup 3 --graphs-per-gpu 4 --gn-stats $gn --cpu-steps 0 --no-exact-fp32 --no-workloads > $OUT/bench_2ranks_1gpu_gloo_$gn.json 2> $OUT/bench_2ranks_1gpu_gloo_$gn.err
  tail -1 $OUT/bench_2ranks_1gpu_gloo_$gn.json | cut -c1-1200
done
