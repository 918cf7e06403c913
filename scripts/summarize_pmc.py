#!/usr/bin/env python
"""Per-kernel averages of every counter found under <dir>/pmc_*/ (rocprofv3 --pmc ... --kernel-trace --output-format csv),
one block per kernel; writes <dir>/pmc_counters.json as well.  FETCH_SIZE is reported raw AND as bytes with the gfx950
half-count correction (x 1024 B x 2); WRITE_SIZE x 1024 B (uncalibrated) - MI355X_MICROARCH.md, HBM section."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("difusco::", "")
        k = k.split("(")[0].replace("void ", "")[:110]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0.0)):
    print(k)
    out[k] = {}
    for c in sorted(acc[k]):
        v = acc[k][c] / cnt[k][c]
        out[k][c] = {"avg_per_launch": v, "launches": cnt[k][c]}
        extra = ""
        if c == "FETCH_SIZE":
            extra = f"   = {v * 1024 * 2 / 1e6:10.1f} MB (x2: gfx950 half-count correction)"
        if c == "WRITE_SIZE":
            extra = f"   = {v * 1024 / 1e6:10.1f} MB"
        if c.startswith("TCC_EA0_RDREQ_") and c.endswith("B_sum"):
            extra = f"   = {v * int(c.split('_')[3][:-1]) / 1e6:10.1f} MB"
        print(f"    {c:34s} {v:18.1f}  ({cnt[k][c]} launches){extra}")
json.dump(out, open(os.path.join(root, "pmc_counters.json"), "w"), indent=1)
