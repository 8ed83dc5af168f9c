#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 3000 python -m pytest tests/test_distributed.py -m gpu -q -k "${1:-gamg}" --durations=5 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -70 ) > $OUT/dist.log 2>&1
tail -50 $OUT/dist.log
