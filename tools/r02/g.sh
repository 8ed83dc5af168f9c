#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -m gpu -x -q 2>&1 | tail -30 ) > $OUT/pytest_gpu_r02g.log 2>&1
tail -6 $OUT/pytest_gpu_r02g.log
