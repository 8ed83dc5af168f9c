#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "restatement or fused_cycle or reuse" 2>&1 | tail -25 ) > $OUT/roots_tests.txt 2>&1
tail -12 $OUT/roots_tests.txt
for v in "PFEM_AMG_ROOTS=1" "PFEM_AMG_ROOTS=0"; do
( env $v timeout 900 python bench.py --workload beam --jitter 0.2 --steps 3 --warmup 1 --no-jacobi-step --no-pmc --no-cpu-baseline 2>$OUT/roots_beam.err | tail -1 ) > $OUT/roots_beam.json
python3 -c "
import json; d=json.load(open('$OUT/roots_beam.json')); p=d['preconditioner']
print('beam jitter [$v]: its', d['iterations'], 'ms', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'rows', p['rows_per_level'], 'sym', round(p['symbolic_setup_ms_once_per_pattern'],1), 'oc', round(p['operator_complexity'],3))" || tail -5 $OUT/roots_beam.err
done
for v in "PFEM_AMG_ROOTS=1" "PFEM_AMG_ROOTS=2"; do
( env $v timeout 900 python bench.py --jitter 0.2 --steps 3 --warmup 1 --no-jacobi-step --no-pmc --no-cpu-baseline 2>$OUT/roots_cube.err | tail -1 ) > $OUT/roots_cube.json
python3 -c "
import json; d=json.load(open('$OUT/roots_cube.json')); p=d['preconditioner']
print('cube jitter [$v]: its', d['iterations'], 'ms', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'rows', p['rows_per_level'], 'sym', round(p['symbolic_setup_ms_once_per_pattern'],1), 'oc', round(p['operator_complexity'],3))" || tail -5 $OUT/roots_cube.err
done
