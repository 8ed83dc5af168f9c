#!/bin/bash
# peer-memory transport between ranks sharing the GPU: its parity cases, then per-exchange latency for 62 KB and 1.3 MB faces
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed.py -x -q -m gpu -k "peer_memory" > $OUT/peer_tests.log 2>&1
tail -25 $OUT/peer_tests.log
