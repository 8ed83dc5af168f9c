#!/bin/bash
# the GPU suite N times in a row on one box (flakiness check at HEAD)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
N=${1:-3}
: > gpurun_out/repeat_suite.log
for i in $(seq 1 $N); do
  r=$(timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -1)
  echo "rep $i: $r" | tee -a gpurun_out/repeat_suite.log
  case "$r" in *failed*|*error*) echo "STOP"; exit 1;; esac
done
