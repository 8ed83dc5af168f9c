#!/bin/bash
# where the dictionary SpMV's time goes: SQ / LDS / cache counters of k_spmvr_vd (separate passes, no tracing domains)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
RX="k_spmvr_vd<false"
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_k
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$RX" -f csv -d /tmp/pmc_k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > /tmp/pmc_k.log 2>&1
  for c in $set; do python3 tools/summarize_prof.py pmc /tmp/pmc_k $c 2>/dev/null | tail -n +2 | head -2 | awk -v c=$c '{print c, $0}' | cut -c1-220; done
done 2>&1 | tee gpurun_out/valdict_pmc.txt
