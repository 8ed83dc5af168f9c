#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
export PFEM_HEAD=${1:-}
timeout 1500 python tools/probe_partition.py ${2:-100} ${3:-3} 2>$OUT/partition.err | grep "^{" | tail -1 > $OUT/partition_${2:-100}_${3:-3}.json
python3 - <<PY
import json
d = json.load(open("$OUT/partition_${2:-100}_${3:-3}.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, {q: v[q] for q in ("iterations", "aggregation", "symbolic_ms_per_rank", "first_solve_ms_gloo_hooks", "warm_solve_ms_gloo_hooks", "distributed_levels", "spmv_rows_per_lane", "spmv_gap_escapes", "value_dictionary_entries")})
        print("   rows", v["rows_per_level_owned_by_rank"][0])
PY
tail -3 $OUT/partition.err
