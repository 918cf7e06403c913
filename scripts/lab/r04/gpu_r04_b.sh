#!/bin/bash
# Round 4, GPU call B: stage-loop lab, phase stamps of the production kernel at two and at one workgroup per CU, full GPU suite.
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
timeout 900 python scripts/bench_stage_lab.py > $OUT/stage_lab.txt 2>&1; echo "lab exit $?" >> $OUT/stage_lab.txt
STAMP_VARIANT=18 timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/20339,17/0" > $OUT/stamps_2wg.txt 2>&1
LDS_PAD=90000 STAMP_VARIANT=18 timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/20339,17/0" > $OUT/stamps_1wg.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -24 $OUT/stage_lab.txt | cut -c1-220
grep -E "median|phase stamps|ticks" $OUT/stamps_2wg.txt | head -24
grep -E "median|phase stamps|ticks" $OUT/stamps_1wg.txt | head -24
grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -30
