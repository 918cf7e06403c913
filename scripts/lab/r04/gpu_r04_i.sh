#!/bin/bash
# Round 4, GPU call I: what the waits / the issue of the full-line gather requests cost (timing-only ablations 34-37, phase stamps).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04i
mkdir -p $OUT
for v in 18 34 35; do
  STAMP_VARIANT=$v timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/151411,36/151411,37/151411" > $OUT/stamps_2wg_$v.txt 2>&1
  LDS_PAD=4000 STAMP_VARIANT=$v timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/151411" > $OUT/stamps_1wg_$v.txt 2>&1
done
for v in 18 34 35; do echo "== two waves per SIMD, stamp variant $v"; grep -E "median|phase stamps|  prologue|  G1|  gather|  LN|  G2 stage 8|  G2 rest|  G2 quarters" $OUT/stamps_2wg_$v.txt; echo "== one wave per SIMD, stamp variant $v"; grep -E "median|phase stamps|  prologue|  G1|  gather|  LN|  G2 stage 8|  G2 rest|  G2 quarters" $OUT/stamps_1wg_$v.txt; done
