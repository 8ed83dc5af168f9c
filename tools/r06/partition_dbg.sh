#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
PFEM_AMG_VERBOSE=1 timeout 1500 python tools/probe_partition.py 60 3 2>$OUT/partition_dbg.err >/dev/null
grep -n "refused\|split\|bricks" $OUT/partition_dbg.err | head -40
