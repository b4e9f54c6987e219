#!/bin/bash
OUT=gpurun_out/sweep; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/pass_a -- python3 tools/sweep_target.py > $OUT/a.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pass_b -- python3 tools/sweep_target.py > $OUT/b.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pass_c -- python3 tools/sweep_target.py > $OUT/c.log 2>&1
python3 tools/sweep_report.py $OUT | tee gpurun_out/sweep_report.txt
