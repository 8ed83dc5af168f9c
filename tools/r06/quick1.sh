#!/bin/bash
# round 6, first lease: the changed paths (value dictionary, assembly-provided bound, barrier-free dot fold, pool) + the bench line with
# its in-run PMC passes
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dictionary or bound_and_diagonal or fused_cycle or codes or spmv" --durations=8 2>&1 | tail -30 ) > $OUT/q1_pytest.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/q1_bench.err | tail -1 ) > $OUT/q1_bench.json
( PFEM_DEBUG_NO_ASM_BOUND=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-jacobi-step 2>$OUT/q1_bench_nobound.err | tail -1 ) > $OUT/q1_bench_nobound.json
tail -5 $OUT/q1_pytest.log
python3 - <<PY
import json
for f in ("q1_bench", "q1_bench_nobound"):
    try:
        d = json.load(open("$OUT/" + f + ".json"))
        r = d["roofline"]
        print(f, d["ms_per_step"], d["iterations"], d["assembly_ms_per_step"], d["preconditioner"]["numeric_setup_ms_per_solve_inside_the_timer"], d["first_step_ms_including_once_per_pattern_setup"],
              "roof", round(r["frac"], 3), round(r["algorithmic_frac"], 3), r["avg_launch_ms"], r["traffic"], (r["traffic_source"] or "")[:60], (d.get("jacobi_step") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/q1_bench.err
