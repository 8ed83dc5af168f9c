#!/bin/bash
# round 3, visit B: first runs of -pc_type gamg
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python tools/probe_amg.py 12 30 60 beam:2 100 beam:4 200 beam:10 2>&1 | tail -40 ) > $OUT/r03b_amg.log 2>&1
cat $OUT/r03b_amg.log | cut -c1-1500
