#!/bin/bash
# same-box A/B of two builds of the library: $1 = alternative .so (in pfemfort_amd/), default build = the other
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do
  for lib in "" "$1"; do
    if [ -n "$lib" ]; then export PFEM_AMD_LIB=$GRAFT_REPO_ROOT/pfemfort_amd/$lib; else unset PFEM_AMD_LIB; fi
    timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-step 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('${lib:-default}', round(d['ms_per_step'], 2), 'ms/step', round(d['ms_per_iteration'] * 1e3, 1), 'us/it', round(d['roofline']['avg_launch_ms'] * 1e3, 1), 'us spmv')"
  done
done
