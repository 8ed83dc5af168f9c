#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q -k "value" 2>&1 | tail -2
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step --no-jacobi-step"
for r in 1 2 3; do
PFEM_VD_VERBOSE=1 timeout 900 python bench.py $F 2>$OUT/cold.err | tail -1 > $OUT/cold.json
python3 -c "
import json; d=json.load(open('$OUT/cold.json'))
print('cfg3 warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2))"
grep "codes refreshed" $OUT/cold.err | head -2
done
PFEM_VD_VERBOSE=1 timeout 900 python bench.py $F --workload beam 2>$OUT/cold.err | tail -1 > $OUT/cold.json
python3 -c "
import json; d=json.load(open('$OUT/cold.json'))
print('beam warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2))"
grep "codes refreshed" $OUT/cold.err | head -2
