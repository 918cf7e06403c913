#!/bin/bash
# Round 4, GPU call Q: the whole GPU suite (incl. the aggregation tests) and smoke on the final tree, as the driver runs them.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04q
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
