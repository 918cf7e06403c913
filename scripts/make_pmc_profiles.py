#!/usr/bin/env python
"""gpurun_out/<dir>/pmc_counters.json (scripts/summarize_pmc.py) -> the tables bench.py reads and the judge can cite:

    profiles/<round>/pmc_traffic.json   "<workload>:<edges>:<variant>" -> kernel -> {fetch_bytes, write_bytes, + L2 / EA level fields}
    profiles/<round>/pmc_sq.json        same key -> {SQ_* per launch of the fused kernel, averaged over the launches of a STEP}
    profiles/<round>/pmc_sq_counters_<workload>.txt    readable per-kernel listing

usage: make_pmc_profiles.py <gpurun_out dir> <round> <workload:edges:variant>"""
import json
import os
import sys

src, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(os.path.join(src, "pmc_counters.json")))
prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", rnd)
os.makedirs(prof, exist_ok=True)


def load(name):
    try:
        return json.load(open(os.path.join(prof, name)))
    except (OSError, ValueError):
        return {}


def val(k, c):
    return d[k][c]["avg_per_launch"] if c in d[k] else None


traffic, sq = load("pmc_traffic.json"), load("pmc_sq.json")
entry = {}
fused = [k for k in d if k.startswith("edge_layer_fused_kernel")]
groups = {}
for k in d:
    short = k.split("<")[0].strip()
    groups.setdefault(short, []).append(k)
for short, ks in groups.items():
    if not any("FETCH_SIZE" in d[k] for k in ks):
        continue
    # launch-weighted average over the instantiations of one kernel (the 12 fused launches of a step: 1 first + 10 middle + 1 last)
    def wavg(c, scale=1.0):
        num = sum(val(k, c) * d[k][c]["launches"] for k in ks if c in d[k])
        den = sum(d[k][c]["launches"] for k in ks if c in d[k])
        return num / den * scale if den else None
    e = {"fetch_bytes": wavg("FETCH_SIZE", 1024 * 2.0), "write_bytes": wavg("WRITE_SIZE", 1024.0)}
    for c, nm, sc in (("TCC_EA0_RDREQ_128B_sum", "ea_read_bytes_128B_requests", 128.0), ("TCC_EA0_RDREQ_64B_sum", "ea_read_bytes_64B_requests", 64.0),
                      ("TCC_EA0_RDREQ_32B_sum", "ea_read_bytes_32B_requests", 32.0), ("TCC_EA0_WRREQ_64B_sum", "ea_write_bytes_64B_requests", 64.0),
                      ("TCC_EA0_RDREQ_DRAM_sum", "ea_read_requests_to_dram", 1.0), ("TCC_EA0_RDREQ_sum", "ea_read_requests", 1.0),
                      ("TCC_READ_sum", "l2_read_requests_128B", 1.0), ("TCC_WRITE_sum", "l2_write_requests", 1.0),
                      ("TCC_HIT_sum", "l2_hits", 1.0), ("TCC_MISS_sum", "l2_misses", 1.0)):
        v = wavg(c, sc)
        if v is not None:
            e[nm] = v
    lv, rq = wavg("TCC_EA0_RDREQ_LEVEL_sum"), wavg("TCC_EA0_RDREQ_sum")
    if lv and rq:
        e["ea_read_latency_cycles"] = lv / rq
    entry[short] = e
traffic[key] = entry
if fused:
    s = {}
    for c in sorted({c for k in fused for c in d[k] if c.startswith("SQ_") or c.startswith("GRBM")}):
        num = sum(val(k, c) * d[k][c]["launches"] for k in fused if c in d[k])
        den = sum(d[k][c]["launches"] for k in fused if c in d[k])
        s[c] = num / den
    sq[key] = s
json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
json.dump(sq, open(os.path.join(prof, "pmc_sq.json"), "w"), indent=1)
wl = key.split(":")[0]
with open(os.path.join(prof, f"pmc_counters_{wl}.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc <set> --kernel-trace, bench.py --steps 2 --warmup 1 (workload {key}); per-launch averages, one pass per\n"
            f"# counter set (scripts/gpu_r04_d.sh); SQ_* in quad-cycles summed over all waves, SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over\n"
            f"# SIMDs, TCC_* summed over the 16 channels x 8 XCDs, GRBM_GUI_ACTIVE summed over the 8 XCDs.  FETCH_SIZE x 1024 x 2 (gfx950\n"
            f"# half-count correction) and WRITE_SIZE x 1024 are the byte figures bench.py reports as `traffic`.\n")
    for k in d:
        if not any(t in k for t in ("edge_layer_fused", "head_apply", "node_linear", "node_finalize", "gn_")):
            continue
        f.write(f"\n{k}\n")
        for c in sorted(d[k]):
            f.write(f"    {c:34s} {d[k][c]['avg_per_launch']:18.1f}   ({d[k][c]['launches']} launches)\n")
print(json.dumps({key: {k: v for k, v in entry.items() if "fused" in k or "head" in k}}, indent=1))
print(json.dumps(sq[key], indent=1) if fused else "")
