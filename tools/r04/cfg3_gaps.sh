#!/bin/bash
# config 3: kernel timeline of the gamg loop -- busy and idle time per iteration, by boundary
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof3
timeout 900 rocprofv3 --kernel-trace -f csv -d /tmp/prof3 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/gaps3.log 2>&1
python tools/trace_gaps.py /tmp/prof3 k_pc_update > $OUT/cfg3_gaps.txt 2>&1
cat $OUT/cfg3_gaps.txt
