#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or any_axis" 2>&1 | tail -30 ) > $OUT/r03d_gamg_tests.log 2>&1
( timeout 300 python tools/probe_amg.py 200 beam:10 2>&1 | cut -c1-900 ) > $OUT/r03d_amg.log 2>&1
tail -30 $OUT/r03d_gamg_tests.log; cat $OUT/r03d_amg.log
