#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
export PFEM_AMG_VERBOSE=1
for P in 10 100; do
  export PFEM_AMG_BRICK_WEAK_PERCENT=$P
  ( timeout 900 python bench.py --workload beam --jitter 0.2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-jacobi-step 2>$OUT/latnum2_b$P.err | tail -1 ) > $OUT/latnum2_b$P.json
  python3 - <<PY
import json
try:
    d=json.load(open("$OUT/latnum2_b$P.json")); print("beam jitter weak% $P", d["ms_per_step"], d["iterations"], d.get("first_step_ms_including_once_per_pattern_setup"), d["preconditioner"].get("rows_per_level"), d["preconditioner"].get("symbolic_setup_ms_once_per_pattern"))
except Exception as e: print("ERR", e)
PY
  grep -a "node bricks" $OUT/latnum2_b$P.err | sort | uniq -c | head
done
