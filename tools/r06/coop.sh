#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "gamg or amg or codes or dictionary or brick or cube_full_size or beam" 2>&1 | tail -25 ) > $OUT/coop_tests.txt 2>&1
tail -6 $OUT/coop_tests.txt
for v in "X=1" "PFEM_CG_COOP=0" "X=2" "PFEM_CG_COOP=0"; do
  ( env $v timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/coop_ab.json
  python3 -c "
import json; d=json.load(open('$OUT/coop_ab.json')); print('bench [$v]', round(d['ms_per_step'],3), d['iterations'], round(d['ms_per_iteration'],4), round(d['assembly_ms_per_step'],3), round(d['solve_ms_per_step'],3), round(d['preconditioner']['numeric_setup_ms_per_solve_inside_the_timer'],3), 'cold', round(d['first_step_ms_including_once_per_pattern_setup'],2), d['max_nodal_error'])"
done
