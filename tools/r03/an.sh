#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export OMP_NUM_THREADS=1
for cfg in "16" "16 -bind-to core -map-by numa" "16 -bind-to numa" "16 -bind-to core:4" "16 -bind-to l3" ; do
  set -- $cfg; n=$1; shift
  /opt/conda/bin/mpiexec -n $n "$@" oracle/pfem_oracle_mpi 200 1e-5 10000 1 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$cfg', 'asm', d['assembly_s'], 'solve', d['solve_s'], 'its', d['iterations'])
    else: print(ln[:160].rstrip())
"
done
