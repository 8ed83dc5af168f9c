#!/bin/bash
# the driver's GPU tier as the driver runs it: pytest -m gpu -x, smoke, bench (tag = $1)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
TAG=${1:-suite}
( timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -150 ) > $OUT/pytest_gpu_$TAG.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $OUT/smoke_$TAG.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench_$TAG.err | tail -1 ) > $OUT/bench_$TAG.json
tail -30 $OUT/pytest_gpu_$TAG.log; cat $OUT/smoke_$TAG.log; cut -c1-600 $OUT/bench_$TAG.json
