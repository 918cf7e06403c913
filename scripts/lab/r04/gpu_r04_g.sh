#!/bin/bash
# Round 4, GPU call G: the whole GPU suite on the final tree, smoke, end-to-end and decode benches, unfused dense bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r04g
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 600 python scripts/bench_end_to_end.py > $OUT/bench_end_to_end.json 2> $OUT/bench_end_to_end.err
timeout 600 python scripts/bench_decode.py > $OUT/bench_decode.json 2> $OUT/bench_decode.err
timeout 300 python bench.py --workload tsp50dense --steps 20 --warmup 5 --no-fusion --cpu-steps 0 --no-exact-fp32 > $OUT/bench_dense_unfused.json 2> $OUT/bench_dense_unfused.err
timeout 300 python bench.py --workload tsp50dense --steps 20 --warmup 5 --cpu-steps 0 --no-exact-fp32 > $OUT/bench_dense_fused.json 2> $OUT/bench_dense_fused.err
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -5 $OUT/bench_end_to_end.json | cut -c1-600; tail -3 $OUT/bench_end_to_end.err
python - <<'PY'
import json
for f in ("bench_dense_unfused", "bench_dense_fused"):
    try:
        o = json.loads(open(f"gpurun_out/r04g/{f}.json").read().strip().splitlines()[-1])
        print(f, round(o["value"], 1), "gs/s", round(o["ms_per_step"], 4), "ms/step")
    except Exception as e:
        print(f, e)
PY
