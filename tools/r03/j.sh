#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 2400 python -m pytest tests/test_bench_contract.py tests/test_fortran_boundary.py tests/test_distributed.py -m gpu -x -q --durations=8 2>&1 | tail -30 ) > $OUT/r03j_tests.log 2>&1
for nb in rcb8 shuffle; do
  ( timeout 900 python bench.py --numbering $nb --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step 2>$OUT/r03j_$nb.err | tail -1 ) > $OUT/r03j_bench_numbering_$nb.json
done
tail -14 $OUT/r03j_tests.log
for nb in rcb8 shuffle; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/r03j_bench_numbering_$nb.json"))
    r=d["roofline"]
    print("$nb", {k:d[k] for k in ("value","ms_per_step","iterations","ms_per_iteration")}, "jacobi", d["jacobi_step"]["ms_per_step"], d["jacobi_step"]["iterations"], d["jacobi_step"]["ms_per_iteration"], "spmv", r["kernel"][:40], r["avg_launch_ms"], r["frac"], r["hbm_frac"], d["setup_s_untimed"])
except Exception as e: print("$nb", "ERR", e); print(open("$OUT/r03j_$nb.err").read()[-1500:])
PY
done
