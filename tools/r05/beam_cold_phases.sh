#!/bin/bash
# config 4 (beam): where the cold step's symbolic phase goes (PFEM_AMG_VERBOSE), plus the distributed gamg tests with the oracle's own bricks
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
PFEM_AMG_VERBOSE=1 timeout 600 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline --no-jacobi-step --no-parity-step > $OUT/beam_verbose.json 2> $OUT/beam_verbose.err
grep "gamg symbolic" $OUT/beam_verbose.err | head -70
tail -1 $OUT/beam_verbose.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','cold_value') if k in d}); print(d.get('cold_step')); print(d.get('amg'))"
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -k "gamg and poisson" -x -q -s 2>&1 | grep -E "bricks across|passed|failed|Error|assert" | head -30
