#!/bin/bash
# Soak: the multi-process GPU tests N times in a row on one box (flakiness check)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-5}
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_distributed.py tests/test_golden_drivers.py tests/test_fortran_boundary.py tests/test_bench_contract.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2 | tr '\n' ' '; echo " [loop $i]"
done 2>&1 | tee gpurun_out/soak.log
