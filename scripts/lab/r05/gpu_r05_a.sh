#!/bin/bash
# Round 5, GPU call A: the dense-workload transient probe + the default bench line of the round-4 kernels on the new bench.py
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 300 python scripts/probe_dense_transient.py > $OUT/probe_dense_transient.txt 2>&1; echo "probe exit $?"
tail -30 $OUT/probe_dense_transient.txt
BENCH_FULL_JSON=$OUT/bench_full.json timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
tail -c 3000 $OUT/bench_default.json
rocm-smi -P -g | head -20
