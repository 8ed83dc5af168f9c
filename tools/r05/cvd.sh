#!/bin/bash
# value codes on the coarse levels: gamg parity cases, then config 3 / config 5 with and without
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or spmv" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "not beam" 2>&1 | tail -3
F="--steps 5 --warmup 2 --no-cpu-baseline --no-parity-step --no-jacobi-step"
for run in 1 2; do
PFEM_VD_VERBOSE=1 timeout 900 python bench.py $F 2>$OUT/cvd.err | tail -1 > $OUT/cvd.json
python3 -c "
import json; d=json.load(open('$OUT/cvd.json')); p=d['preconditioner']
print('cfg3 its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'ms/it', round(d['ms_per_iteration'],4), 'dicts', p['value_dictionary_entries_per_level'], 'rnorm', d['rnorm'])"
done
grep "level" $OUT/cvd.err | sort | uniq -c | head
PFEM_VD_VERBOSE=1 timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>$OUT/cvd5.err | tail -1 > $OUT/cvd.json
python3 -c "
import json; d=json.load(open('$OUT/cvd.json')); p=d['preconditioner']
print('cfg5 its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'dicts', p['value_dictionary_entries_per_level'], 'rnorm', d['rnorm'])"
grep "level" $OUT/cvd5.err | sort | uniq -c | head
