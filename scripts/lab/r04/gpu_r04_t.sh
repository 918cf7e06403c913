#!/bin/bash
# Round 4, GPU call T: the stage-loop geometries of stage_lab_all_variants.txt again, this time in the POWER-LIMITED STEADY STATE
# (3 s of back-to-back launches per case, two alternating rounds, rocm-smi beside each) - the burst timings of bench_stage_lab.py
# are taken before the governor settles, and the two can rank differently (stage_lab_mfma_16x16x32.txt).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04t
mkdir -p $OUT
timeout 600 python scripts/bench_lab_power.py 142020,143120,242020,244120,182020,184120,16142020 800000 3 > $OUT/lab_geometries_steady.txt 2> $OUT/lab_geometries_steady.err
tail -2 $OUT/lab_geometries_steady.err
grep -v "^{" $OUT/lab_geometries_steady.txt
