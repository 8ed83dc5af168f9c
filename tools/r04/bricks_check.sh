#!/bin/bash
# bricks in one step: gamg parity cases, then config 3's phase timings and line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gamg" > $OUT/bricks_tests.log 2>&1
tail -5 $OUT/bricks_tests.log
for BR in 1 0; do
( PFEM_AMG_BRICKS=$BR PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>$OUT/bricks_cfg3_$BR.err | tail -1 ) > $OUT/bricks_cfg3_$BR.json
grep -E "gamg symbolic level [01] " $OUT/bricks_cfg3_$BR.err | head -40
python3 - <<PY
import json
d=json.load(open("$OUT/bricks_cfg3_$BR.json")); p=d["preconditioner"]
print("bricks=$BR", {k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, p["rows_per_level"], p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
PY
done
