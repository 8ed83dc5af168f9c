#!/bin/bash
# coupled gamg hierarchy at size: iteration counts against the number of ranks (ranks share the one GPU), weak scaling
# workload (200^3 per rank along z) and the beam on 8 ranks
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r03ab_ranks.log
for N in 2 4 8; do
  ( timeout 1200 python bench.py --gpus $N --same-device --backend gloo --steps 1 --warmup 1 --no-strong-block --no-jacobi-step 2>$OUT/r03ab_$N.err | tail -1 ) > $OUT/r03ab_bench_$N.json
  python3 - <<PY >> $OUT/r03ab_ranks.log
import json
try:
    d=json.load(open("$OUT/r03ab_bench_$N.json"))
    print($N, d["config"]["free_dofs"], "gamg its", d["iterations"], "ms/step", round(d["ms_per_step"],1), "levels", d["preconditioner"].get("rows_per_level"), d["preconditioner"].get("form"))
except Exception as e: print($N, "ERR", e)
PY
done
( timeout 900 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 0 --no-jacobi-step 2>$OUT/r03ab_beam.err | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam 8 ranks', d['iterations'], round(d['ms_per_step'],1), d['preconditioner'].get('rows_per_level'))" ) >> $OUT/r03ab_ranks.log 2>&1
cat $OUT/r03ab_ranks.log
tail -3 $OUT/r03ab_8.err
