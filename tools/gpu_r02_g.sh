#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1500 python -m pytest tests/test_distributed.py tests/test_bench_contract.py tests/test_golden_drivers.py -m gpu -x -q 2>&1 | tail -30 ) > $OUT/pytest_gpu_r02g.log 2>&1
( timeout 600 python tools/probe_overlap.py 200 200 2>$OUT/probe_overlap.err | grep '^{' | tail -1 ) > $OUT/probe_overlap_200.json
( timeout 600 python tools/probe_overlap.py 100 200 2>>$OUT/probe_overlap.err | grep '^{' | tail -1 ) > $OUT/probe_overlap_100.json
tail -5 $OUT/pytest_gpu_r02g.log; cat $OUT/probe_overlap_200.json $OUT/probe_overlap_100.json; tail -5 $OUT/probe_overlap.err
