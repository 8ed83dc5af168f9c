#!/bin/bash
# gamg parity cases, the config-4 beam bench line and its kernel trace
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gamg" > $OUT/beam_check_tests.log 2>&1
tail -5 $OUT/beam_check_tests.log
( timeout 900 python bench.py --workload beam --steps 3 --warmup 1 --no-jacobi-step 2>$OUT/beam_check.err | tail -1 ) > $OUT/beam_check.json
python3 - <<PY
import json
d=json.load(open("$OUT/beam_check.json")); p=d["preconditioner"]
print({k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, p["rows_per_level"], p["numeric_setup_ms_per_solve_inside_the_timer"], p["symbolic_setup_ms_once_per_pattern"])
PY
rm -rf /tmp/prof_beam
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_beam -- python3 bench.py --workload beam --steps 3 --warmup 1 --no-jacobi-step > $OUT/prof_beam.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_beam 32 > $OUT/prof_beam_kernel_stats.txt 2>&1
cat $OUT/prof_beam_kernel_stats.txt
