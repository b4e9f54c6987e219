#!/bin/bash
# PMC comparison of stream-kernel variants: bash tools/prof_pmc.sh <tag> "<SPMV_TUNING string>"
TAG=$1; export SPMV_TUNING="$2"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum --output-format csv -d $OUT/pmc_tcp -- $BENCH > $OUT/tcp.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/pmc_ta -- $BENCH > $OUT/ta.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
python3 tools/prof_summary.py $OUT > gpurun_out/pmc_${TAG}_summary.txt
