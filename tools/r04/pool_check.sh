#!/bin/bash
# the device block pool: config 5 set-up and cold step in one process (bench), repeated pattern builds, then the whole GPU suite
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for P in 64 0; do
( PFEM_POOL_GB=$P timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/pool_cfg5_$P.json
python3 - <<PY
import json
d=json.load(open("$OUT/pool_cfg5_$P.json")); p=d["preconditioner"]
print("pool_gb=$P", {k:d.get(k) for k in ("value","ms_per_step","iterations","first_step_ms_including_once_per_pattern_setup","setup_s_untimed")}, p["symbolic_setup_ms_once_per_pattern"], d["setup_breakdown_s"], d["device_memory_gb"])
PY
done
( timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $OUT/pool_suite.log 2>&1
cat $OUT/pool_suite.log
