#!/bin/bash
# Galerkin maps of a brick level from the member lists (default) against the 32-bit sort (PFEM_AMG_BRICK_SORT=1):
# symbolic phases and cold step at config 3, then the gamg tests
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for S in 0 1; do
  if [ $S = 1 ]; then export PFEM_AMG_BRICK_SORT=1; else unset PFEM_AMG_BRICK_SORT; fi
  ( PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>$OUT/brick_maps_$S.err | tail -1 ) > $OUT/brick_maps_$S.json
  grep -E "gamg symbolic" $OUT/brick_maps_$S.err | head -12
  python3 - <<PY
import json
d=json.load(open("$OUT/brick_maps_$S.json")); p=d["preconditioner"]
print("sort=$S", {k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","first_step_ms_including_once_per_pattern_setup")}, p["rows_per_level"], p["nnz_per_level"], p["gershgorin_lambda_max"][:3], p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
PY
done
unset PFEM_AMG_BRICK_SORT
( timeout 2400 python -m pytest tests -m gpu -x -q -k "gamg or amg or full_size or bricks or coupled" 2>&1 | tail -5 ) > $OUT/brick_maps_tests.log 2>&1
cat $OUT/brick_maps_tests.log
