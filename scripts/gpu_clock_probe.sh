#!/bin/bash
# effective shader clock of the fused edge-layer kernel = GRBM_GUI_ACTIVE / kernel duration, for 2 and 1 workgroups per CU
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; cd /tmp
for pad in 0 12288; do
  LDS_PAD=$pad timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $REPO/gpurun_out/clk_$pad -o p -- python $REPO/scripts/bench_fused_layer.py fp16x3 0 nostamp > $REPO/gpurun_out/clk_$pad.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob
for pad in (0, 12288):
    cc = glob.glob(f"gpurun_out/clk_{pad}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"gpurun_out/clk_{pad}/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(pad, "missing output"); continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if "edge_layer_fused" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    vals = []
    for r in csv.DictReader(open(cc[0])):
        if "edge_layer_fused" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            vals.append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    vals = vals[len(vals) // 2:]
    if vals:
        cyc = sum(v[0] for v in vals) / len(vals); ns = sum(v[1] for v in vals) / len(vals)
        print(f"LDS pad {pad}: {len(vals)} launches, GUI_ACTIVE {cyc:.0f} cycles, duration {ns/1e3:.1f} us -> {cyc/ns:.3f} GHz")
PY
