#!/bin/bash
# block-Jacobi/gamg iteration counts against the number of ranks (ranks share the one GPU: counts are what matters here)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r03l_blocks.log
for N in 2 4 8; do
  ( timeout 1200 python bench.py --gpus $N --same-device --backend gloo --steps 1 --warmup 1 --no-strong-block 2>$OUT/r03l_$N.err | tail -1 ) > $OUT/r03l_bench_$N.json
  python3 - <<PY >> $OUT/r03l_blocks.log
import json
try:
    d=json.load(open("$OUT/r03l_bench_$N.json"))
    print($N, d["config"]["free_dofs"], "gamg its", d["iterations"], "ms/step", round(d["ms_per_step"],1), "jacobi its", d["jacobi_step"]["iterations"], "ms/step", round(d["jacobi_step"]["ms_per_step"],1), "levels", d["preconditioner"]["rows_per_level"])
except Exception as e: print($N, "ERR", e)
PY
done
cat $OUT/r03l_blocks.log
