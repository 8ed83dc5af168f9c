#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "$1" 2>&1 | tail -40 ) > $OUT/t_parity.txt 2>&1
tail -40 $OUT/t_parity.txt
