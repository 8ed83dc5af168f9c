#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "dictionary or codes or bound_and or cube_full_size or reuse or parity" 2>&1 | tail -25 ) > $OUT/direct_tests.txt 2>&1
tail -6 $OUT/direct_tests.txt
for v in "X=1" "PFEM_VD_DIRECT_OFF=1"; do
  ( env $v timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 ) > $OUT/direct_ab.json
  python3 -c "
import json; d=json.load(open('$OUT/direct_ab.json')); print('bench [$v]', round(d['ms_per_step'],3), d['iterations'], round(d['ms_per_iteration'],4), round(d['assembly_ms_per_step'],3), round(d['solve_ms_per_step'],3), round(d['preconditioner']['numeric_setup_ms_per_solve_inside_the_timer'],3), 'cold', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'jacobi', round(d['jacobi_step']['ms_per_step'],2))"
done
