#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
N=${1:-300}
timeout 3000 python tools/r05/scatter_soak_inprocess.py $N 2>&1 | tee gpurun_out/scatter_soak_inprocess.log | tail -20
