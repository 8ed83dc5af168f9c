#!/bin/bash
# where a slow symbolic phase goes: hipMalloc time and pool hits per phase (PFEM_POOL_VERBOSE), right behind a multi-process pytest
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
for i in 1 2; do
PFEM_POOL_VERBOSE=1 timeout 600 python bench.py $F 2>$OUT/ap_$i.err | tail -1 > $OUT/ap_$i.json
python3 -c "
import json; d=json.load(open('$OUT/ap_$i.json'))
print('run $i', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['setup_breakdown_s'].items()})"
grep -E "pool:|gamg symbolic phase" $OUT/ap_$i.err | head -6
done
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -1
for i in 3 4; do
PFEM_POOL_VERBOSE=1 timeout 600 python bench.py $F 2>$OUT/ap_$i.err | tail -1 > $OUT/ap_$i.json
python3 -c "
import json; d=json.load(open('$OUT/ap_$i.json'))
print('run $i (behind pytest)', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['setup_breakdown_s'].items()})"
grep -E "pool:|gamg symbolic phase" $OUT/ap_$i.err | head -6
done
