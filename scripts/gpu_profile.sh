#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box through gpurun).  Writes under gpurun_out/prof_*;
# scripts/summarize_prof.py condenses it into gpurun_out/prof_summary_*.txt for profiles/.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
TAG=${1:-r03}
KEY=${2:-tsp1000:800000:fused-fp16x3}
EXTRA=${PROF_BENCH_ARGS:-}
BENCH_ARGS="--steps 5 --warmup 2 --cpu-steps 0 --no-profile --no-exact-fp32 $EXTRA"
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats_$TAG -o bench -- python $REPO/bench.py $BENCH_ARGS > $REPO/gpurun_out/prof_stats_$TAG.log 2>&1
echo "stats exit $?" >> $REPO/gpurun_out/prof_stats_$TAG.log
PMC_ARGS="--steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 $EXTRA"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_fetch_$TAG -o bench -- python $REPO/bench.py $PMC_ARGS > $REPO/gpurun_out/prof_fetch_$TAG.log 2>&1
echo "fetch exit $?" >> $REPO/gpurun_out/prof_fetch_$TAG.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_write_$TAG -o bench -- python $REPO/bench.py $PMC_ARGS > $REPO/gpurun_out/prof_write_$TAG.log 2>&1
echo "write exit $?" >> $REPO/gpurun_out/prof_write_$TAG.log
cd $REPO
python scripts/summarize_prof.py $TAG $KEY > gpurun_out/prof_summary_${TAG}_${KEY//:/_}.txt 2>&1
cat gpurun_out/prof_summary_${TAG}_${KEY//:/_}.txt
# keep the merged-back payload small: drop the raw per-dispatch traces, keep stats + counters
find gpurun_out/prof_stats_$TAG -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/prof_* | tail -8
