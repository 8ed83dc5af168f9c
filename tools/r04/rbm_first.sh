#!/bin/bash
# round 4: first run of the rigid-body-mode coarse space -- gamg parity cases, then the config-4 beam with and without it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gamg" > $OUT/rbm_first_tests.log 2>&1
tail -15 $OUT/rbm_first_tests.log
for RBM in 1 0; do
  ( PFEM_AMG_RBM=$RBM PFEM_AMG_VERBOSE=1 timeout 900 python bench.py --workload beam --steps 3 --warmup 1 2>$OUT/rbm_first_beam_rbm$RBM.err | tail -1 ) > $OUT/rbm_first_beam_rbm$RBM.json
  python3 - <<PY
import json
try:
    d=json.load(open("$OUT/rbm_first_beam_rbm$RBM.json"))
    print("rbm=$RBM", {k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, d["preconditioner"])
except Exception as e: print("rbm=$RBM ERR", e)
PY
  grep -E "gamg symbolic" $OUT/rbm_first_beam_rbm$RBM.err | head -60
done
