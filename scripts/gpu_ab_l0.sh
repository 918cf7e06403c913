#!/bin/bash
# A/B of the layer-0 table fold on bench.py workloads (same process conditions, alternating)
for rep in 1 2; do
for w in tsp1000 mis; do
  for flag in "" "--no-l0-fold"; do
    python bench.py --workload $w --steps 20 --warmup 3 --cpu-steps 0 --profile-all $flag 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w $flag', round(d['value'],1), 'g-s/s', round(d['ms_per_step'],3), 'ms/step', {k:round(v['ms_total']/d['steps'],3) for k,v in d['kernels'].items() if isinstance(v,dict) and "ms_total" in v})"
  done
done
done
