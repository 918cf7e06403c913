#!/bin/bash
# SQ-level PMC passes on bench.py (kernel-trace only; never combined with sys/hip traces).
# usage: gpurun -- 'bash scripts/gpu_pmc.sh TAG "COUNTER LIST 1" "COUNTER LIST 2" ...'
cd $GRAFT_REPO_ROOT
REPO=$PWD
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out
cd /tmp
[ -f $REPO/gpurun_out/counters.txt ] || rocprofv3 -L > $REPO/gpurun_out/counters.txt 2>&1
i=0
for SET in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$i -o bench -- \
    python $REPO/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-profile ${BENCH_EXTRA} > $REPO/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc set $i ($SET) exit $?"
done
cd $REPO
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("difusco::", "")[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "fused" in k or "gate" in k or "linear_rows" in k:
                print(k, {c: round(v / cnt[(k, c)], 1) for c, v in acc[k].items()})
PY
