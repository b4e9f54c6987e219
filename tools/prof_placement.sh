#!/bin/bash
# rocprofv3 --pmc passes over tools/placement_pmc.py (one group per pass, each under its own timeout).
# Usage (GPU box): bash tools/prof_placement.sh <tag>
OUT=gpurun_out/placement_pmc_$1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -o -i "\b[A-Z0-9_]*\(UTCL2\|TLB\|MALL\|DRAM\|HBM\|EA0_RD\|EA0_WR\|ATC\|XNACK\|FRAG\)[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_names.txt || true
i=0
while read -r group; do
  i=$((i + 1))
  timeout -k 10 200 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_$i -- python3 tools/placement_pmc.py 10 > $OUT/pmc_$i.log 2>&1 || { echo "pass $i ($group) FAILED"; tail -3 $OUT/pmc_$i.log; continue; }
  echo "== pass $i: $group"
  python3 tools/placement_pmc_report.py $OUT/pmc_$i $OUT/pmc_$i.log | tee $OUT/report_$i.txt
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_TAG_STALL_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
GROUPS
