#!/bin/bash
# where the cold first step goes at config 3 (and config 5 with $1 = 400): PFEM_AMG_VERBOSE phase timings + the bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
N=${1:-200}
( PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --cells $N --steps 3 --warmup 1 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>$OUT/cold_phases_$N.err | tail -1 ) > $OUT/cold_phases_$N.json
grep -E "gamg symbolic" $OUT/cold_phases_$N.err | head -40
python3 - <<PY
import json
d=json.load(open("$OUT/cold_phases_$N.json")); p=d["preconditioner"]
print({k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","first_step_ms_including_once_per_pattern_setup","setup_breakdown_s")}, p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
PY
