#!/bin/bash
# what bounds the assembly kernels: SQ counters of k_gather_poisson_tet4 (default bench) and k_gather_elast_rows (beam)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
PASS1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
PASS2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
PASS3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
: > $OUT/gather_sq_counters.txt
for W in cube beam; do
  ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step"
  [ $W = beam ] && ARGS="--workload beam $ARGS"
  i=0
  for P in "$PASS1" "$PASS2" "$PASS3"; do
    i=$((i+1))
    rm -rf /tmp/prof_sq
    timeout 600 rocprofv3 --pmc $P --kernel-include-regex "k_gather" -f csv -d /tmp/prof_sq -- python3 bench.py $ARGS > $OUT/gather_sq_${W}_$i.log 2>&1
    for C in $P; do
      echo "== $W $C" >> $OUT/gather_sq_counters.txt
      python tools/summarize_prof.py pmc /tmp/prof_sq $C | tail -n +2 >> $OUT/gather_sq_counters.txt 2>&1
    done
  done
done
cat $OUT/gather_sq_counters.txt | cut -c1-60,90-140
