#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rbm or rigid or elast or cook or gamg" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k beam 2>&1 | tail -2
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg and elast" 2>&1 | tail -2
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam"
for sortenv in 0 1 0 1; do
if [ $sortenv = 1 ]; then export PFEM_AMG_BRICK_SORT=1; else unset PFEM_AMG_BRICK_SORT; fi
timeout 900 python bench.py $F 2>/dev/null | tail -1 > $OUT/q2.json
python3 -c "
import json; d=json.load(open('$OUT/q2.json'))
print('beam sorted_maps=$sortenv', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), d['iterations'], d['rnorm'])"
done
unset PFEM_AMG_BRICK_SORT
timeout 1500 python -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -2
