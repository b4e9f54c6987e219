#!/bin/bash
OUT=gpurun_out/sweep; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
PASSES=${SWEEP_PASSES:-"FETCH_SIZE,TCC_HIT_sum TCP_TCC_READ_REQ_sum,TA_BUSY_avr,TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum,TCC_MISS_sum,TCC_REQ_sum"}
i=0
for p in $PASSES; do
  i=$((i+1))
  rocprofv3 --pmc ${p//,/ } --output-format csv -d $OUT/pass_$i -- python3 tools/sweep_target.py > $OUT/$i.log 2>&1
done
python3 tools/sweep_report.py $OUT | tee gpurun_out/sweep_report.txt
