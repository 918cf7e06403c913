#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the TSP-10000 Gaussian step's kernels (one PMC pass each): the table embedding kernel as a write stream
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; OUT=gpurun_out/r06_pmc10k; mkdir -p $OUT; cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/$OUT/pmc_$C -o bench -- python $REPO/bench.py --workload tsp10000 --steps 2 --warmup 1 --cpu-steps 0 --no-profile --no-exact-fp32 --repeats 1 --no-power > $REPO/$OUT/pmc_$C.log 2>&1
  echo "$C exit $?"
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = r["Kernel_Name"].replace("difusco::", "")[:60]
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in acc:
            res[k][c] = acc[k] / cnt[k] * 1024 * (2.0 if c == "FETCH_SIZE" else 1.0)
            res[k]["n"] = cnt[k]
with open(out + "/pmc_traffic_tsp10000.txt", "w") as fh:
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:10]:
        line = "%-62s launches %4d  fetch %9.1f MB  write %9.1f MB per launch" % (k, v.get("n", 0), v.get("FETCH_SIZE", 0) / 1e6, v.get("WRITE_SIZE", 0) / 1e6)
        print(line); fh.write(line + "\n")
PY
