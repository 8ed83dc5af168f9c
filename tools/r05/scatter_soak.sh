#!/bin/bash
# soak the two files in suite order (parity, then the multi-GB full-size tests) with the block pool and without it;
# test_elasticity_beam_config4 now compares the atomic-scatter K with the ORACLE and dumps the differing entries on a mismatch
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
LOG=$OUT/scatter_soak.log
: > $LOG
timeout 900 python -m pytest tests/test_distributed.py -m gpu -k "peer" -x -q 2>&1 | tail -3 | tee -a $LOG
N=${1:-10}
for pool in default 0; do
  for i in $(seq 1 $N); do
    t0=$(date +%s)
    if [ "$pool" = "0" ]; then export PFEM_POOL_GB=0; else unset PFEM_POOL_GB; fi
    r=$(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -1)
    t1=$(date +%s)
    echo "pool=$pool rep=$i $((t1-t0))s: $r" | tee -a $LOG
    case "$r" in *failed*|*error*) echo "STOP: failure" | tee -a $LOG; ls $OUT/scatter_mismatch_* 2>/dev/null | tee -a $LOG; exit 1;; esac
  done
done
echo "all clean" | tee -a $LOG
