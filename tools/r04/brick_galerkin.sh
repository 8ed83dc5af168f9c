#!/bin/bash
# numeric Galerkin product of a brick level by coarse row (k_lat_galerkin, default) against the map-driven kernel
# (PFEM_AMG_GALERKIN_MAPS=1): step time and numeric set-up at config 3, bit-equality of the solve, then the gamg tests
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for S in 0 1 0 1; do
  if [ $S = 1 ]; then export PFEM_AMG_GALERKIN_MAPS=1; else unset PFEM_AMG_GALERKIN_MAPS; fi
  ( timeout 900 python bench.py --steps 20 --warmup 3 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>/dev/null | tail -1 ) > $OUT/brick_gal_$S.json
  python3 - <<PY
import json
d=json.load(open("$OUT/brick_gal_$S.json")); p=d["preconditioner"]
print("maps=$S", {k:d.get(k) for k in ("ms_per_step","iterations","rnorm","first_step_ms_including_once_per_pattern_setup")}, p["gershgorin_lambda_max"][:4], p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
PY
done
unset PFEM_AMG_GALERKIN_MAPS
( timeout 2400 python -m pytest tests -m gpu -x -q -k "gamg or amg or full_size or bricks or coupled" 2>&1 | tail -5 ) > $OUT/brick_gal_tests.log 2>&1
cat $OUT/brick_gal_tests.log
