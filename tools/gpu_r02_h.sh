#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for ov in 1 0; do for n in 200 100; do
( PFEM_MULTI_OVERLAP=$ov timeout 300 python tools/probe_overlap.py $n 200 2>$OUT/probe_ov.err | grep '^{' | tail -1 ) > $OUT/probe_overlap_${n}_ov$ov.json
echo "overlap=$ov n=$n"; cat $OUT/probe_overlap_${n}_ov$ov.json
done; done
