#!/bin/bash
# bricks of three nodes per axis on level 0 of displacement problems: parity cases, then config 4 with 2 / 3 / 4
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step --no-jacobi-step --workload beam"
for b in 4 5 6 8 4; do
PFEM_AMG_NODE_BRICK0=$b timeout 900 python bench.py $F 2>/dev/null | tail -1 > $OUT/b3.json
python3 -c "
import json; d=json.load(open('$OUT/b3.json')); p=d['preconditioner']
print('beam first bricks $b: its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'ms/it', round(d['ms_per_iteration'],3), 'rows', p['rows_per_level'], 'nnz', p['nnz_per_level'][:3], 'tip', d.get('max_displacement'))"
done
