#!/bin/bash
# config 3 (200^3): where the cold first step goes (PFEM_AMG_VERBOSE phase timings), the driver-form line, kernel trace
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-jacobi-step --no-cpu-baseline --no-parity-step 2>$OUT/cfg3_phases.err | tail -1 ) > $OUT/cfg3_phases.json
grep -E "gamg symbolic|pattern|ms" $OUT/cfg3_phases.err | head -120
python3 - <<PY
import json
d=json.load(open("$OUT/cfg3_phases.json")); p=d["preconditioner"]
print({k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, p["rows_per_level"], p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
print({k:v for k,v in d.items() if "ms" in k or "setup" in k})
PY
rm -rf /tmp/prof3
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof3 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/prof3.log 2>&1
python tools/summarize_prof.py stats /tmp/prof3 60 > $OUT/cfg3_kernel_stats.txt 2>&1
cat $OUT/cfg3_kernel_stats.txt
