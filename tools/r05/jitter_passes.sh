#!/bin/bash
# the beam with moved nodes: more pairing passes on level 0 (aggregates of 16 / 32 / 64 nodes under the rigid-body coarse space)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam --jitter 0.2"
for p in 3 4 5 6; do
PFEM_AMG_PASSES0=$p timeout 900 python bench.py $F 2>/dev/null | tail -1 > $OUT/jp.json
python3 -c "
import json; d=json.load(open('$OUT/jp.json')); p=d['preconditioner']
print('moved beam, level-0 passes $p: its', d['iterations'], 'warm', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'rows', p['rows_per_level'][:4], 'complexity', round(p['operator_complexity'],2))"
done
