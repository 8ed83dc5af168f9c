#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( PFEM_AMG_VERBOSE=1 timeout 1500 python bench.py --gpus 8 --same-device --backend gloo --steps 1 --warmup 0 --no-transport-ab --no-jacobi-step --no-strong-block 2>$OUT/dbg8.err | tail -1 ) > $OUT/dbg8.json
python3 -c "
import json; c=json.load(open('$OUT/dbg8.json')); p=c['preconditioner']
print('cfg5 x8', c['iterations'], p['rows_per_level'], p.get('distributed_levels'), c['ms_per_step'])"
grep "bricks refused" $OUT/dbg8.err | sort | uniq -c | head
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -3
( timeout 900 python bench.py --gpus 4 --same-device --backend gloo --steps 1 --warmup 0 --cells 100 --no-transport-ab --no-jacobi-step --no-strong-block 2>/dev/null | tail -1 ) > $OUT/dbg4.json
python3 -c "
import json; c=json.load(open('$OUT/dbg4.json')); p=c['preconditioner']
print('100^3/rank x4', c['iterations'], p['rows_per_level'], p.get('distributed_levels'))"
