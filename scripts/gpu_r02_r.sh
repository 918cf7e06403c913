#!/bin/bash
# phase stamps: what do the B h[i] gathers (one or two distinct rows per tile) and the A / V gathers cost?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r; mkdir -p $O
for V in 18 30 31 24; do
  STAMP_VARIANT=$V timeout 200 python scripts/bench_fused_layer.py fp16x3 "0/883" > $O/stamps_$V.log 2>&1; echo "== stamps variant $V"; grep -A9 "phase stamps" $O/stamps_$V.log
done
