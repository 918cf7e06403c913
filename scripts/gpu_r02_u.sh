#!/bin/bash
# SQ wait / activity counters of the fused middle-layer kernel: production (OPT 3955) beside all options off (OPT 0)
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out/u; cd /tmp
for V in 3955 0; do
  i=0
  for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $REPO/gpurun_out/u/pmc_${V}_$i -o p -- \
      python $REPO/scripts/bench_fused_layer.py fp16x3 "0/$V" nostamp > $REPO/gpurun_out/u/pmc_${V}_$i.log 2>&1
    echo "opt $V set $i exit $?"
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for V in (3955, 0):
    out = {}
    for d in sorted(glob.glob(f"gpurun_out/u/pmc_{V}_*/")):
        for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(float); cnt = collections.Counter()
            for r in csv.DictReader(open(f)):
                if "edge_layer_fused" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            out.update({c: round(v / cnt[c]) for c, v in acc.items()})
    print("OPT", V, out)
PY
