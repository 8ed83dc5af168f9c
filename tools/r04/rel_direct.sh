#!/bin/bash
# assembly writes the SpMV's relative-group copy itself: parity tests that multiply / solve after an assembly, full-size cubes, bench A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu > $OUT/rel_direct_tests.log 2>&1
tail -5 $OUT/rel_direct_tests.log
for D in 0 1; do
  E=""; [ $D = 0 ] && E="PFEM_DEBUG_NO_REL_DIRECT=1"
  ( env $E A=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/rel_direct_$D.json
  python3 - <<PY
import json
d=json.load(open("$OUT/rel_direct_$D.json")); p=d["preconditioner"]
print("direct=$D", {k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","assembly_ms_per_step","solve_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, p["numeric_setup_ms_per_solve_inside_the_timer"], d["max_nodal_error"])
PY
done
