#!/bin/bash
# config 4 on 8 ranks sharing the GPU (gloo hooks: counts, not times): passes on level 0 of the hierarchy across ranks
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for p in 3 4 5; do
PFEM_AMG_PASSES0=$p timeout 900 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step --no-parity-step 2>$OUT/b8.err | tail -1 > $OUT/b8.json
python3 -c "
import json; d=json.load(open('$OUT/b8.json')); p=d['preconditioner']
print('beam on 8 ranks, level-0 passes $p: its', d['iterations'], 'reason', d['converged_reason'], 'ms', round(d['ms_per_step'],1), 'rows', p['rows_per_level'][:4], 'tip', d.get('check') or d.get('tip_displacement'))" || tail -3 $OUT/b8.err
done
