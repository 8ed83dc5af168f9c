#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/pytest_parity_r02f.log 2>&1
for c in poisson100 poisson200 beam; do
  ( timeout 900 python tools/probe_ilu0_gpu.py $c 2>$OUT/ilu0_gpu_$c.err | tail -1 ) > $OUT/ilu0_gpu_$c.json
done
tail -4 $OUT/pytest_parity_r02f.log; cat $OUT/ilu0_gpu_*.json; tail -3 $OUT/ilu0_gpu_*.err
