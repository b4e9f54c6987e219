#!/bin/bash
# Memory-system counters (L2 = TCC, L1 = TCP) for one program, one rocprofv3 --pmc pass per counter group (a group
# the hardware cannot collect together makes rocprofv3 abort and linger: every pass runs under its own timeout).
# Run on the GPU box:  bash tools/prof_memsys.sh <out dir under gpurun_out> python3 <script> [args]
# Summarise with tools/prof_summary.py <out dir>.
OUT=gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
while read -r group; do
  i=$((i + 1))
  timeout -k 10 150 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_$i -- "$@" > $OUT/pmc_$i.log 2>&1 || { echo "pass $i ($group) FAILED"; exit 1; }
  echo "pass $i ($group) ok"
done <<'GROUPS'
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_STREAMING_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN2_sum
TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TOTAL_ACCESSES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_REQUEST_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
GROUPS
