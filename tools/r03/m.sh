#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "renumbering" 2>&1 | tail -30 ) > $OUT/r03m_tests.log 2>&1
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "reorder or gamg" 2>&1 | tail -12 ) >> $OUT/r03m_tests.log 2>&1
for nb in shuffle rcb8; do
  ( timeout 900 python bench.py --numbering $nb --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step 2>$OUT/r03m_$nb.err | tail -1 ) > $OUT/r03m_bench_numbering_$nb.json
done
( PFEM_REORDER=1 timeout 900 python bench.py --numbering rcb8 --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step 2>$OUT/r03m_rcb8f.err | tail -1 ) > $OUT/r03m_bench_numbering_rcb8_forced.json
tail -30 $OUT/r03m_tests.log
for nb in shuffle rcb8 rcb8_forced; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/r03m_bench_numbering_$nb.json"))
    r=d["roofline"]
    print("$nb", {k:d[k] for k in ("value","ms_per_step","iterations","ms_per_iteration")}, "jacobi", d["jacobi_step"]["ms_per_step"], d["jacobi_step"]["iterations"], "spmv", r["kernel"][:34], round(r["avg_launch_ms"],4), round(r["frac"],3), "setup", round(d["setup_s_untimed"],2))
except Exception as e: print("$nb", "ERR", e)
PY
done
