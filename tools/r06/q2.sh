#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for V in 1 0; do
  ( PFEM_INC_PATTERNS=$V timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-pmc --no-parity-step 2>$OUT/q2_$V.err | tail -1 ) > $OUT/q2_$V.json
  python3 - <<PY
import json
d=json.load(open("$OUT/q2_$V.json")); print("patterns=$V", d["ms_per_step"], d["iterations"], d.get("assembly_ms_per_step"), d["setup_breakdown_s"])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "incidence or numbering" 2>&1 | tail -2
