#!/bin/bash
# Round 4, GPU call R: GEMM 1 alone on the 16x16x32 MFMA shape against the production 32x32x16 shape (same scheme, same data),
# interleaved rounds; then ~5 s of each beside rocm-smi.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04r
mkdir -p $OUT
timeout 600 python scripts/bench_stage_lab.py 142020n,16142020n,16142021n,142020n,16142020n > $OUT/lab16.txt 2> $OUT/lab16.err
tail -3 $OUT/lab16.err
grep -v "^{" $OUT/lab16.txt
timeout 300 python scripts/bench_lab_power.py 16142020 > $OUT/lab16_power.txt 2> $OUT/lab16_power.err
tail -2 $OUT/lab16_power.err
grep -v "^{" $OUT/lab16_power.txt
