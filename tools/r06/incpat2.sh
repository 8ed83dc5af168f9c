#!/bin/bash
# round 6: HBM traffic of the gather kernel per dispatch, pattern table against the nodes' own records
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "incidence" 2>&1 | tail -5 ) > $OUT/incpat2_parity.txt 2>&1
tail -3 $OUT/incpat2_parity.txt
: > $OUT/incpat2_pmc.txt
for V in pat own; do
  if [ $V = own ]; then export PFEM_INC_PATTERNS=0; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_ip$C
    timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_gather" -f csv -d /tmp/prof_ip$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/incpat2_pmc_$C.log 2>&1
    echo "== $V" >> $OUT/incpat2_pmc.txt
    python tools/summarize_prof.py pmc_each /tmp/prof_ip$C $C k_gather >> $OUT/incpat2_pmc.txt 2>&1
  done
done
cat $OUT/incpat2_pmc.txt
