#!/bin/bash
# rocprofv3 passes for the default bench workload.  Run on the GPU box:
#   bash tools/prof.sh <tag> [extra bench args]
# Writes gpurun_out/prof_<tag>/{trace,pmc_*}/...; copy the summaries you want judged into profiles/.
set -e
TAG=${1:-r2}; shift || true
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
# (upload's placement search launches the kernel a few dozen times more: off here, so that the per-product division of
# the counters in tools/prof_traffic.py stays exact; the JSON line of every pass -- trace.log, pmc_*.log -- carries the
# box record and the placement the pass ran on)
# (the same for upload's with / without timing of the pattern plan: here the plan is simply kept -- what the un-profiled
# bench decides for this workload on every box seen, config.pattern_plan in its line)
export SPMV_TUNING="${SPMV_TUNING:+$SPMV_TUNING,}place_tries=0,local_patterns=1"
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
# counters in their own passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
if [ -n "$PROF_LIGHT" ]; then find $OUT -name "*.csv" | head -10; exit 0; fi
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- $BENCH > $OUT/pmc_tcp.log 2>&1 || true
find $OUT -name "*.csv" | head -40
