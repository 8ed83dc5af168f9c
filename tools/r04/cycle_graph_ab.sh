#!/bin/bash
# the gamg cycle replayed from a hipGraph (default) against stream launches (PFEM_CG_GRAPH=0): warm and cold step at 200^3, 100^3, 50^3
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for N in 200 100 50; do
for G in 1 0 1 0; do
  ( PFEM_CG_GRAPH=$G timeout 600 python bench.py --cells $N --steps 20 --warmup 3 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>/dev/null | tail -1 ) > $OUT/cg.json
  python3 - <<PY
import json
d=json.load(open("$OUT/cg.json")); p=d["preconditioner"]
print("n=$N graph=$G", {k:round(d.get(k),3) for k in ("ms_per_step","first_step_ms_including_once_per_pattern_setup")}, d["iterations"], round(p["ms_per_cycle_with_its_cg_iteration"],4), round(p["symbolic_setup_ms_once_per_pattern"],2))
PY
done
done
