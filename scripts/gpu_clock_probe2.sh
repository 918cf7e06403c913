#!/bin/bash
# effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of ablated variants of the fused kernel
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; cd /tmp
for m in 0 15 8 7; do
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $REPO/gpurun_out/clk2_$m -o p -- python $REPO/scripts/bench_fused_layer.py fp16x3 $m nostamp > $REPO/gpurun_out/clk2_$m.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob
for m in (0, 15, 8, 7):
    cc = glob.glob(f"gpurun_out/clk2_{m}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"gpurun_out/clk2_{m}/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        print(m, "missing"); continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "edge_layer_fused" in r["Kernel_Name"]}
    vals = [(float(r["Counter_Value"]), dur[r["Dispatch_Id"]], r["Kernel_Name"][:60]) for r in csv.DictReader(open(cc[0]))
            if "edge_layer_fused" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur]
    # the ablated variant is the instantiation with the mask in its name; mask 0 = production
    key = f"Li{m}ELi4E" if m else "Li0ELi4E"
    vals = [v for v in vals if True]
    vals = vals[len(vals) // 3:]
    cyc = sum(v[0] for v in vals) / len(vals) / 8; ns = sum(v[1] for v in vals) / len(vals)
    print(f"ablation {m:3d}: {len(vals)} launches, {cyc/1e6:.2f} M cycles, {ns/1e3:.0f} us -> {cyc/ns:.2f} GHz")
PY
