#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "gamg or amg or codes or dictionary or brick or cube_full_size" 2>&1 | tail -25 ) > $OUT/gal_tests.txt 2>&1
tail -6 $OUT/gal_tests.txt
for v in "X=1" "PFEM_AMG_GALERKIN_EXTRAS=0"; do
  ( env $v timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/gal_ab.json
  python3 -c "
import json; d=json.load(open('$OUT/gal_ab.json')); print('bench [$v]', round(d['ms_per_step'],3), d['iterations'], round(d['ms_per_iteration'],4), round(d['assembly_ms_per_step'],3), round(d['solve_ms_per_step'],3), round(d['preconditioner']['numeric_setup_ms_per_solve_inside_the_timer'],3), 'cold', round(d['first_step_ms_including_once_per_pattern_setup'],2))"
done
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/gal_prof.log 2>&1
python tools/trace_phase.py /tmp/prof_stats > $OUT/gal_phase.txt 2>&1
head -34 $OUT/gal_phase.txt
