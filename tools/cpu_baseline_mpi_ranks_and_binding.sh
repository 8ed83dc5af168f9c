#!/bin/bash
# MPI CPU baseline: ranks and binding
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export OMP_NUM_THREADS=1
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -c 300; echo
python3 -c "import os; print(len(os.sched_getaffinity(0)))"
for cfg in "16 -bind-to core" "32 -bind-to core" "64 -bind-to core" "64" "128" "64 -bind-to hwthread" "128 -bind-to numa"; do
  set -- $cfg; n=$1; shift
  /opt/conda/bin/mpiexec -n $n "$@" oracle/pfem_oracle_mpi 200 1e-5 10000 1 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('$cfg', 'asm', d['assembly_s'], 'solve', d['solve_s'], 'its', d['iterations'])
    else: print(ln[:200].rstrip())
"
done
