#!/bin/bash
# config 5 alone on one GPU: where the cold first step goes (three fresh processes: the stack's allocation stalls come and go)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3; do
( PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>$OUT/cfg5_cold_$i.err | tail -1 ) > $OUT/cfg5_cold_$i.json
grep -E "gamg symbolic level [-01] " $OUT/cfg5_cold_$i.err | head -12
python3 - <<PY
import json
d=json.load(open("$OUT/cfg5_cold_$i.json")); p=d["preconditioner"]
print("run $i", {k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, p["symbolic_setup_ms_once_per_pattern"], d["setup_breakdown_s"])
PY
done
