#!/bin/bash
# Round 4, GPU call M: is the matrix-core rate set by socket power?  GEMM 1 alone on N(0,1) data vs all-zero data (same instruction
# stream, same bytes), rocm-smi sampled beside each (scripts/bench_lab_power.py).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04m
mkdir -p $OUT
timeout 600 python scripts/bench_lab_power.py 142020 > $OUT/lab_power.txt 2> $OUT/lab_power.err
tail -3 $OUT/lab_power.err
grep -v "^{" $OUT/lab_power.txt
