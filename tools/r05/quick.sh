#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or amg or rbm or brick" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -2
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
for w in poisson beam; do
for i in 1 2; do
timeout 600 python bench.py --workload $w $F 2>/dev/null | tail -1 > $OUT/q_$w$i.json
python3 -c "
import json; d=json.load(open('$OUT/q_$w$i.json'))
print('$w', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), d['iterations'])"
done; done
