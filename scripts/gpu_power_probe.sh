#!/bin/bash
# socket power and shader clock while bench.py runs (rocm-smi polled every ~0.25 s)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 1500 --warmup 5 --cpu-steps 0 --no-profile > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*sclk clock level: [0-9S]*: (\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' '; echo
done > gpurun_out/power_samples.txt
wait $BP
sort -t' ' -k4 -n gpurun_out/power_samples.txt | tail -8
python -c "
import json; d=json.load(open('gpurun_out/power_bench.json')); print('bench', round(d['value'],1), 'graph-steps/s', round(d['ms_per_step'],3), 'ms/step')"
