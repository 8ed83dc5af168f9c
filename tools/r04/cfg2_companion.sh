#!/bin/bash
# the Jacobi companion step of the default bench at config 2 under different pool settings
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for V in "" "" "" "PFEM_POOL_GB=0" ""; do
  ( env $V timeout 600 python bench.py --cells 100 --steps 5 --warmup 2 --no-cpu-baseline --no-parity-step 2>/dev/null | tail -1 ) > $OUT/c2c.json
  python3 - <<PY
import json
d=json.load(open("$OUT/c2c.json")); j=d["jacobi_step"]
print("[$V]", round(d["ms_per_step"],3), "jacobi:", {k:(round(j.get(k),3) if isinstance(j.get(k),float) else j.get(k)) for k in ("ms_per_step","iterations","ms_per_iteration")})
PY
done
