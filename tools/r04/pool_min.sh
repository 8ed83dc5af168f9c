#!/bin/bash
# smallest pooled block: cold step and symbolic phase at config 3 for PFEM_POOL_MIN_KB = 32768 (32 MiB) / 1024 / 64 / 4
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for K in 32768 1024 64 4 32768 64; do
  ( PFEM_POOL_MIN_KB=$K timeout 600 python bench.py --steps 10 --warmup 2 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>/dev/null | tail -1 ) > $OUT/pool_min_$K.json
  python3 - <<PY
import json
d=json.load(open("$OUT/pool_min_$K.json")); p=d["preconditioner"]
print("min_kb=$K", {k:d.get(k) for k in ("ms_per_step","first_step_ms_including_once_per_pattern_setup","setup_s_untimed")}, p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"], d["setup_breakdown_s"]["symbolic_pattern_and_incidence"], d["setup_breakdown_s"]["symbolic_pattern_and_incidence_second_build"])
PY
done
