#!/bin/bash
# elasticity assembly writing the SpMV's node-group copy itself: parity + full-size beam tests, bench A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "elast or beam or group or row_forms or renumbering or gamg or cook" > $OUT/grp_direct_tests.log 2>&1
tail -4 $OUT/grp_direct_tests.log
for D in 0 1; do
  E=""; [ $D = 0 ] && E="PFEM_DEBUG_NO_REL_DIRECT=1"
  ( env $E A=1 timeout 900 python bench.py --workload beam --steps 5 --warmup 2 --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/grp_direct_$D.json
  python3 - <<PY
import json
d=json.load(open("$OUT/grp_direct_$D.json")); p=d["preconditioner"]
print("direct=$D", {k:d.get(k) for k in ("value","ms_per_step","iterations","assembly_ms_per_step","solve_ms_per_step")}, p["numeric_setup_ms_per_solve_inside_the_timer"], d.get("max_displacement_magnitude"))
PY
done
