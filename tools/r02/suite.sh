#!/bin/bash
# GPU visit: the -m gpu suite as the driver runs it (optionally twice), then whatever extra command is given
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r02}
REPS=${2:-1}
for i in $(seq 1 $REPS); do
  ( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -60 ) > $OUT/pytest_gpu_${TAG}_$i.log 2>&1
  tail -12 $OUT/pytest_gpu_${TAG}_$i.log
done
