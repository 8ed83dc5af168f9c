#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for W in 2 8; do
timeout 600 python tools/probe_peer.py $W 2>$OUT/probe_peer_$W.err | grep "^{" | tail -1 > $OUT/probe_peer_$W.json
cat $OUT/probe_peer_$W.json; echo
done
tail -3 $OUT/probe_peer_2.err
