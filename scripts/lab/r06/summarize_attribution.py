"""Per-variant averages of the PMC passes of scripts/lab/r06/gpu_b.sh: one row per instantiation of edge_layer_fused_kernel (the ABL
template argument names the access class that is switched off)."""
import collections
import csv
import glob
import re
import sys

src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(src + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "edge_layer_fused_kernel" not in k:
            continue
        m = re.search(r"edge_layer_fused_kernel<([^>]*)>", k)
        key = m.group(1) if m else k
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
LABEL = {"0": "production", "3072": "no A h[j] / V h[j] gather requests", "128": "no B h[i] loads", "3200": "no node-table access at all",
         "16384": "no weight-stage refills", "32768": "no e stream into GEMM 1", "32": "no residual read, no e store"}
print("# per launch averages (E = 800,000, N = 8,000: TSP-1000 x 8, fp16x3 middle layer); FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024")
rows = []
for key in acc:
    a = {c: v / cnt[(key, c)] for c, v in acc[key].items()}
    abl = key.split(",")[1].strip() if "," in key else "?"
    rows.append((abl, key, a))
base = next((a for abl, _, a in rows if abl == "0"), None)
for abl, key, a in sorted(rows, key=lambda r: int(r[0]) if r[0].isdigit() else 1 << 30):
    fetch = a.get("FETCH_SIZE", 0) * 1024 * 2
    write = a.get("WRITE_SIZE", 0) * 1024
    line = f"ABL {abl:>6s} {LABEL.get(abl, ''):38s} fetch {fetch / 1e9:7.3f} GB  write {write / 1e9:7.3f} GB"
    if base:
        line += f"  d(fetch) {(fetch - base.get('FETCH_SIZE', 0) * 2048) / 1e9:+7.3f} GB"
    for c in ("TCC_READ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
              "TCC_EA0_RDREQ_DRAM_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCC_WRITE_sum"):
        if c in a:
            line += f"  {c.replace('_sum', '')} {a[c] / 1e6:8.3f} M"
    print(line)
    print(f"           <{key}> launches {max(cnt[(key, c)] for c in a)}")
