#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for pr in 0; do
( PFEM_MULTI_OVERLAP=$pr timeout 300 python tools/probe_overlap.py 200 200 2>$OUT/probe_ov.err | grep '^{' | tail -1 ) > $OUT/probe_overlap_200_ov1_prio$pr.json
echo "overlap=default(by size) run"; cat $OUT/probe_overlap_200_ov1_prio$pr.json
done
