#!/bin/bash
# rigid-body modes in the hierarchy across the ranks: the multi-rank parity cases (ranks share the GPU, gloo hooks), then the
# config-4 beam on 8 ranks sharing the GPU (iteration count against the one-GPU hierarchy)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_distributed.py -x -q -m gpu -k "gamg or elast" -s > $OUT/coupled_rbm_tests.log 2>&1
grep -E "^gamg |passed|failed|Error|error" $OUT/coupled_rbm_tests.log | tail -40
tail -5 $OUT/coupled_rbm_tests.log
( timeout 1500 python bench.py --workload beam --gpus 8 --same-device --backend gloo --steps 1 --warmup 1 --no-jacobi-step 2>$OUT/coupled_rbm_beam8.err | tail -1 ) > $OUT/coupled_rbm_beam8.json
python3 - <<PY
import json
try:
    d=json.load(open("$OUT/coupled_rbm_beam8.json")); p=d["preconditioner"]
    print({k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration")}, p.get("rows_per_level"), p.get("distributed_levels"), p.get("communication_per_cycle"))
except Exception as e: print("ERR", e); print(open("$OUT/coupled_rbm_beam8.err").read()[-3000:])
PY
