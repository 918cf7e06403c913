#!/bin/bash
# final build: 2 ranks sharing the one GPU over gloo (plumbing), end-to-end and decode benches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/t; mkdir -p $O
timeout 300 python scripts/bench_end_to_end.py 2>/dev/null | grep '^{' > $O/bench_end_to_end.json; cut -c1-600 $O/bench_end_to_end.json
timeout 300 python scripts/bench_decode.py 2>/dev/null | grep '^{' > $O/bench_decode.json; cut -c1-200 $O/bench_decode.json
export HSA_ENABLE_IPC_MODE_LEGACY=0 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo
for GN in per_shard_call global; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --graphs-per-gpu 4 --gn-stats $GN 2> $O/bench_2rank_gloo_$GN.err | grep '^{' > $O/bench_2rank_gloo_$GN.json
echo "2 ranks on one GPU ($GN) exit $?"; python -c "import json; d=json.load(open('$O/bench_2rank_gloo_$GN.json')); print(d['n_gpus'], round(d['value'],1), d['config']['gn_stats'], d['config']['global_batch'])"
done
