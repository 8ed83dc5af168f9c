#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for c in "SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
  rm -rf /tmp/prof_c
  timeout 600 rocprofv3 --pmc $c --kernel-include-regex "k_gather" -f csv -d /tmp/prof_c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step > $OUT/prof_c.log 2>&1
  for k in $c; do python tools/summarize_prof.py pmc /tmp/prof_c $k | tail -1; done
done
