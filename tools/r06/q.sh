#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  ( timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-jacobi-step --no-pmc --no-parity-step 2>$OUT/q_$i.err | tail -1 ) > $OUT/q_$i.json
  python3 - <<PY
import json
d=json.load(open("$OUT/q_$i.json")); print("run $i", d["ms_per_step"], d["iterations"], d.get("assembly_ms_per_step"))
PY
done
