#!/bin/bash
# r06 call B: WHICH accesses are the 0.85 GB of re-reads per launch of the fused kernel.
# One process runs the production kernel and the one-access-class-off ablations (their kernel names differ by the ABL template
# argument), one rocprofv3 PMC pass per counter set (FETCH_SIZE and WRITE_SIZE never share a pass).
cd $GRAFT_REPO_ROOT; REPO=$PWD; export TMPDIR=/tmp; OUT=$REPO/gpurun_out/r06b; mkdir -p $OUT; cd /tmp
MASKS="0,37,41,42,38,39,40"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- \
    python $REPO/scripts/bench_fused_layer.py fp16x3 $MASKS nostamp > $OUT/pmc_$i.log 2>&1
  echo "set $i ($SET) exit $?" >> $OUT/sets.txt
done
cd $REPO
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python scripts/lab/r06/summarize_attribution.py $OUT > $OUT/reread_attribution.txt 2>&1
cat $OUT/sets.txt; cat $OUT/reread_attribution.txt; grep "median" $OUT/pmc_1.log
