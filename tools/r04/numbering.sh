#!/bin/bash
# config 3's mesh under the reference's 8-part RCB renumbering and under a random node permutation: which SpMV form, how fast
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
for NB in rcb8 shuffle; do
( timeout 1200 python bench.py --numbering $NB --steps 5 --warmup 2 --no-cpu-baseline --no-parity-step 2>$OUT/numbering_$NB.err | tail -1 ) > $OUT/numbering_$NB.json
python3 - <<PY
import json
d=json.load(open("$OUT/numbering_$NB.json")); r=d["roofline"]
print("$NB", {k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, "jacobi", (d.get("jacobi_step") or {}).get("ms_per_step"), r["kernel"][:70], round(r["frac"],3), round(r["avg_launch_ms"],4), d["preconditioner"]["rows_per_level"], d["preconditioner"]["levels_paired_on_the_lattice"])
PY
done
