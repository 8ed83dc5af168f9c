#!/bin/bash
# how closely a pooled block must fit the request: config 5 set-up and cold step for PFEM_POOL_FIT = 2 / 4 / 16
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for F in 2 4 16 2 4; do
  ( PFEM_POOL_FIT=$F PFEM_POOL_VERBOSE=1 timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>$OUT/pool_fit_$F.err | tail -1 ) > $OUT/pool_fit_$F.json
  grep "pool:" $OUT/pool_fit_$F.err | head -2
  python3 - <<PY
import json
d=json.load(open("$OUT/pool_fit_$F.json")); p=d["preconditioner"]
print("fit=$F", {k:d.get(k) for k in ("ms_per_step","first_step_ms_including_once_per_pattern_setup","setup_s_untimed")}, p["symbolic_setup_ms_once_per_pattern"], {k:round(v,3) for k,v in d["setup_breakdown_s"].items() if "pattern" in k}, d["device_memory_gb"])
PY
done
