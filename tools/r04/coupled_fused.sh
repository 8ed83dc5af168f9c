#!/bin/bash
# coupled cycle: fused pack / unpack against separate kernels (bits), all multi-rank gamg cases, then the self-peer probe
# (RCCL, the rank as its own neighbour) with the bottom replicated from 32768 / 150000 / 1000000 rows
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_distributed.py -x -q -m gpu -k "fused or gamg" > $OUT/coupled_fused_tests.log 2>&1
tail -4 $OUT/coupled_fused_tests.log
for F in 1 0; do
  PFEM_AMG_COUPLED_FUSED=$F timeout 600 python tools/probe_coupled.py 200 30 > $OUT/probe_coupled_fused$F.json 2>$OUT/probe_coupled_fused$F.err
  python3 - <<PY
import json
d=json.load(open("$OUT/probe_coupled_fused$F.json"))
for k in ("one_rank_loop","coupled_replicated_bottom","coupled_all_levels_distributed"):
    r=d[k]; print("fused=$F", k, {q:r[q] for q in ("ms_per_iteration","host_enqueue_ms_per_iteration","host_ms_inside_rccl_calls_per_iteration","levels","distributed_levels","numeric_setup_ms")})
PY
done
