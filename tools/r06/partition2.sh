#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
export PFEM_HEAD=${1:-}
timeout 1500 python tools/probe_partition.py ${2:-160} ${3:-3} 2>$OUT/partition.err | grep "^{" | tail -1 > $OUT/final_partition_stairs_${2:-160}cube_${3:-3}ranks.json
python3 -c "
import json
d=json.load(open('$OUT/final_partition_stairs_${2:-160}cube_${3:-3}ranks.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, v['iterations'], v['aggregation'], v['symbolic_ms_per_rank'], v['rows_per_level_owned_by_rank'][0], v['first_solve_ms_gloo_hooks'], v['warm_solve_ms_gloo_hooks'])
"
