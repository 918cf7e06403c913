#!/usr/bin/env python
"""Condense rocprofv3 output of scripts/gpu_profile.sh: per-kernel time table (from --stats) and
per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB-class
units of the TCC EA request counters; FETCH_SIZE reads exactly half of the bytes of wide coalesced
streaming reads on gfx950 -> the read side is doubled.  WRITE_SIZE is uncalibrated (reported as is)."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# key of the profiled run in profiles/<round>/pmc_traffic.json: "<workload>:<edges on rank 0>:<variant>" (bench.py looks its
# own run up under exactly this key and reports traffic = null when the run was never profiled)
run_key = sys.argv[2] if len(sys.argv) > 2 else "tsp1000:800000:fused-fp16x3"
root = "gpurun_out"


def find(d, pat):
    hits = glob.glob(os.path.join(root, d, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("difusco::", "")
    return name[:90]


stats = find(f"prof_stats_{tag}", "*kernel_stats.csv")
print(f"# rocprofv3 --kernel-trace --stats  (bench.py --steps 5 --warmup 2 {os.environ.get('PROF_BENCH_ARGS', '')}; run key {run_key})")
if stats:
    rows = list(csv.DictReader(open(stats)))
    print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:14]:
        print(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
else:
    print("no kernel_stats.csv found")

traffic = {}
for counter, d in (("FETCH_SIZE", f"prof_fetch_{tag}"), ("WRITE_SIZE", f"prof_write_{tag}")):
    f = find(d, "*counter_collection.csv")
    print(f"\n# rocprofv3 --pmc {counter}  (per launch average, bench.py --steps 2 --warmup 1)")
    if not f:
        print("no counter_collection.csv found")
        continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    print(f"{'kernel':92s} {'launches':>8s} {'raw_avg':>14s} {'MB_per_launch':>14s}")
    for k in sorted(acc, key=lambda k: -acc[k])[:10]:
        raw = acc[k] / cnt[k]
        mb = raw * 1024 / 1e6 * (2.0 if counter == "FETCH_SIZE" else 1.0)
        print(f"{k:92s} {cnt[k]:8d} {raw:14.1f} {mb:14.1f}")
        short_key = k.split("(")[0].replace("void ", "").split("<")[0].strip()
        # template instantiations of one kernel share an entry: launch-weighted average
        t = traffic.setdefault(short_key, {"fetch_bytes": 0.0, "write_bytes": 0.0, "_n": {}})
        field = "fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"
        n0 = t["_n"].get(field, 0)
        t[field] = (t[field] * n0 + mb * 1e6 * cnt[k]) / (n0 + cnt[k])
        t["_n"][field] = n0 + cnt[k]
    print("(MB_per_launch = raw * 1024 B" + (" * 2 (gfx950 FETCH_SIZE half-count correction)" if counter == "FETCH_SIZE" else " (uncalibrated)") + ")")

import json
for t in traffic.values():
    t.pop("_n", None)
path = os.path.join(root, f"pmc_traffic_{tag}.json")
try:
    table = json.load(open(path))
except (OSError, ValueError):
    table = {}
table[run_key] = traffic
json.dump(table, open(path, "w"), indent=1)
