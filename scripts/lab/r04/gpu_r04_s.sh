#!/bin/bash
# Round 4, GPU call S: GEMM 1 alone, 32x32x16 against 16x16x32 MFMA shape, SAME box, power-limited steady state (5 s per case,
# three alternating rounds, rocm-smi beside each).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04s
mkdir -p $OUT
timeout 600 python scripts/bench_lab_power.py 142020,16142020 > $OUT/lab_shapes_power.txt 2> $OUT/lab_shapes_power.err
tail -2 $OUT/lab_shapes_power.err
grep -v "^{" $OUT/lab_shapes_power.txt
