#!/bin/bash
# GPU visit: the -m gpu suite (twice, as the driver runs it) and the root-cause probe of the round-1 hang
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
TAG=${1:-r02a}
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -40 ) > $OUT/pytest_gpu_${TAG}_1.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_gpu_${TAG}_2.log 2>&1
if [ "${2:-}" = "probe" ]; then
  rm -rf $OUT/gloo_probe
  ( timeout 900 python tools/probe_gloo_device_hang.py 30 null 2>&1 | tail -30 ) > $OUT/gloo_probe_null.log 2>&1
  ( timeout 600 python tools/probe_gloo_device_hang.py 15 own 2>&1 | tail -30 ) > $OUT/gloo_probe_own.log 2>&1
fi
tail -8 $OUT/pytest_gpu_${TAG}_1.log; tail -3 $OUT/pytest_gpu_${TAG}_2.log; cat $OUT/gloo_probe/summary.txt 2>/dev/null | tail -20
