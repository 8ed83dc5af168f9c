#!/bin/bash
# the coupled gamg loop in its single-reduction form: parity cases, then the self-peer probe with and without it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_distributed.py -x -q -m gpu -k "gamg_single or (single and not peer)" -s > $OUT/gamg_single_tests.log 2>&1
grep -E "^gamg |passed|failed|Error" $OUT/gamg_single_tests.log | tail -12
for S in 0 1; do
  PFEM_CG_SINGLE_REDUCTION=$S timeout 600 python tools/probe_coupled.py 200 30 2>$OUT/gamg_single_probe$S.err | grep "^{" | tail -1 > $OUT/gamg_single_probe$S.json
  python3 - <<PY
import json
d=json.load(open("$OUT/gamg_single_probe$S.json"))
for k in ("one_rank_loop","coupled_replicated_bottom"):
    r=d[k]; print("single=$S", k, {q:(round(r[q],4) if isinstance(r[q],float) else r[q]) for q in ("ms_per_iteration","host_enqueue_ms_per_iteration","iterations","reason")})
PY
done
