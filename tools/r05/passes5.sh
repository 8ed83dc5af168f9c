#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_distributed.py -m gpu -x -q -k "elast or rbm or beam" 2>&1 | tail -3
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam"
for a in "--jitter 0.2" "--numbering shuffle" ""; do
timeout 900 python bench.py $F $a 2>/dev/null | tail -1 > $OUT/p5.json
python3 -c "
import json; d=json.load(open('$OUT/p5.json')); p=d['preconditioner']
print('beam $a: its', d['iterations'], 'warm', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'rows', p['rows_per_level'][:4], 'complexity', round(p['operator_complexity'],2))"
done
timeout 900 python bench.py --mode compat --cells 30 2>/dev/null | tail -1 | cut -c1-300
