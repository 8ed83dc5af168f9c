#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for ov in 0 1; do
  ( PFEM_MULTI_OVERLAP=$ov timeout 300 python tools/probe_overlap.py 400 200 50 2>$OUT/probe_slab.err | grep '^{' | tail -1 ) > $OUT/probe_cfg5slab_overlap$ov.json
  echo "overlap=$ov"; cat $OUT/probe_cfg5slab_overlap$ov.json; echo
done
