#!/bin/bash
# round 6: a lattice by numbering (nodes moved off their sites): --jitter lines with and without it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for V in 1 0; do
  export PFEM_AMG_LATTICE_BY_NUMBERING=$V
  ( timeout 900 python bench.py --jitter 0.2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>$OUT/latnum_p$V.err | tail -1 ) > $OUT/latnum_p$V.json
  ( timeout 900 python bench.py --workload beam --jitter 0.2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-jacobi-step 2>$OUT/latnum_b$V.err | tail -1 ) > $OUT/latnum_b$V.json
  for f in latnum_p$V latnum_b$V; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); print("$f", d["ms_per_step"], d["iterations"], d.get("first_step_ms_including_once_per_pattern_setup"), d["preconditioner"].get("rows_per_level"), d["preconditioner"].get("aggregation"), d["preconditioner"].get("symbolic_setup_ms_once_per_pattern"))
except Exception as e: print("$f ERR", e)
PY
  done
done
tail -3 $OUT/latnum_p1.err $OUT/latnum_b1.err
