#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests/test_gpu_full_size.py tests/test_distributed.py tests/test_fortran_boundary.py tests/test_fortran_drivers.py tests/test_bench_contract.py tests/test_abi.py -m gpu -q --durations=8 2>&1 | tail -60 ) > $OUT/pytest_rest.log 2>&1
tail -40 $OUT/pytest_rest.log
