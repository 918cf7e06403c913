#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
for V in 18 27 28 29 23; do
  STAMP_VARIANT=$V timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/115" > $O/stamps_$V.log 2>&1; echo "== stamps variant $V"; grep -A9 "phase stamps" $O/stamps_$V.log
done
