#!/bin/bash
# round 2, GPU call G: full GPU suite on the new default kernel, default bench, rocprofv3 stats + PMC for the headline run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 300 python scripts/bench_fused_layer.py fp16x3 "0/0,0/19" nostamp 2>&1 | grep -v amdgpu.ids | tee $O/fused_ab.log
bash scripts/gpu_profile.sh r02 tsp1000:800000:fused-fp16x3 > $O/profile.log 2>&1; tail -45 $O/profile.log
