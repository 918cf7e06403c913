#!/bin/bash
# A/B of the GroupNorm-partials fold on bench.py (alternating runs)
for rep in 1 2; do
  for flag in "" "--no-gn-fold"; do
    python bench.py --workload tsp1000 --steps 20 --warmup 3 --cpu-steps 0 --profile-all $flag 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tsp1000 $flag', round(d['value'],1), 'g-s/s', round(d['ms_per_step'],3), 'ms/step', {k:round(v['ms_total']/d['steps'],3) for k,v in d['kernels'].items() if isinstance(v,dict) and "ms_total" in v})"
  done
done
