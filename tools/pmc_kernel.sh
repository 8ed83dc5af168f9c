#!/bin/bash
# PMC counters for kernels matching a regex (separate passes, no tracing domains): tools/pmc_kernel.sh <regex> <script...>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
RX=$1; shift
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_k
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$RX" -f csv -d /tmp/pmc_k -- python3 "$@" > /tmp/pmc_k.log 2>&1
  for c in $set; do python3 tools/summarize_prof.py pmc /tmp/pmc_k $c 2>/dev/null | tail -n +2 | head -3 | awk -v c=$c '{print c, $0}' | cut -c1-200; done
done
