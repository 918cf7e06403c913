#!/bin/bash
# end of round 2: full GPU suite, smoke, default bench with the final library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json; cut -c1-260 $O/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 2>/dev/null | grep '^{' > $O/bench_20.json
python -c "import json; r=json.load(open('$O/bench_20.json')); print('20 steps', r['value'], r['ms_per_step'], r['roofline']['frac'])"
