#!/bin/bash
# bench.py at the three arithmetic modes of the E-row linears
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for P in ${PRECS:-bf16x6 fp16x3 bf16x3 fp32}; do
  timeout 600 python bench.py --steps 10 --warmup 2 --cpu-steps 0 --precision $P > gpurun_out/bench_$P.json 2> gpurun_out/bench_$P.err
  echo "== $P exit $?"; cat gpurun_out/bench_$P.json
done
