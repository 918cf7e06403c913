#!/bin/bash
# round 2, GPU call H: torch custom ops test, per-workload bench lines, 64-graph single-GPU line, 2 ranks on one GPU, oracle thread scan
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 300 python -m pytest tests -m gpu -q -x -k "torch_custom_ops or sequential or dense_merge" 2>&1 | tail -3
python -c "import psutil, os; print('cpu: physical', psutil.cpu_count(logical=False), 'logical', os.cpu_count())"
timeout 400 python scripts/cpu_threads_scan.py 16 32 64 128 2>&1 | tee $O/cpu_threads_scan.txt
for w in tsp500 tsp10000 mis; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --cpu-steps 0 > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "import json; d=json.load(open('$O/bench_$w.json')); r=d['roofline']; print('$w', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step frac', round(r['frac'],3), 'fused ms', round(r['avg_launch_ms'],4), 'exact_fp32', round(d['exact_fp32']['value'],1))"
done
timeout 600 python bench.py --graphs-per-gpu 64 --steps 5 --warmup 1 --cpu-steps 0 > $O/bench_tsp1000_64graphs.json 2> $O/bench_tsp1000_64graphs.err
python -c "import json; d=json.load(open('$O/bench_tsp1000_64graphs.json')); r=d['roofline']; print('tsp1000 x64 graphs', round(d['value'],1), 'gs/s', round(d['ms_per_step'],3), 'ms/step frac', round(r['frac'],3), 'fused ms', round(r['avg_launch_ms'],4))"
export HSA_ENABLE_IPC_MODE_LEGACY=0 BENCH_SINGLE_DEVICE=1 BENCH_BACKEND=gloo
for GN in per_shard_call global; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 5 --warmup 2 --graphs-per-gpu 4 --gn-stats $GN > $O/bench_2rank_gloo_$GN.json 2> $O/bench_2rank_gloo_$GN.err
echo "2 ranks on one GPU ($GN) exit $?"; python -c "import json; d=json.load(open('$O/bench_2rank_gloo_$GN.json')); print(d['n_gpus'], round(d['value'],1), d['config']['gn_stats'], d['config']['global_batch'])"
done
