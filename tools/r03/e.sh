#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -30 ) > $OUT/r03e_dist_gamg.log 2>&1
( timeout 600 python bench.py --steps 10 --warmup 3 2>$OUT/r03e_bench.err | tail -1 ) > $OUT/r03e_bench_n1.json
( timeout 600 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step 2>>$OUT/r03e_bench.err | tail -1 ) > $OUT/r03e_bench_cfg5_gamg.json
( timeout 600 python bench.py --workload beam --steps 2 --warmup 1 2>>$OUT/r03e_bench.err | tail -1 ) > $OUT/r03e_bench_beam.json
( timeout 900 python bench.py --gpus 2 --same-device --backend gloo --cells 100 --steps 2 --warmup 1 2>>$OUT/r03e_bench.err | tail -1 ) > $OUT/r03e_bench_2ranks.json
tail -12 $OUT/r03e_dist_gamg.log; tail -5 $OUT/r03e_bench.err
for f in n1 cfg5_gamg beam 2ranks; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/r03e_bench_$f.json"))
    print("$f", {k:d[k] for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step")}, d.get("jacobi_step"), d["preconditioner"].get("rows_per_level"), d["roofline"]["frac"], d.get("strong_cfg5",{}).get("ms_per_step"))
except Exception as e: print("$f", "ERR", e)
PY
done
