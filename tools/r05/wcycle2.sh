#!/bin/bash
# W-cycle: how deep do the second visits have to go?  (PFEM_AMG_W_TO = last level visited twice)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
for wl in "beam 0.2" "poisson 0.2" "beam 0" "poisson 0"; do
set -- $wl
G="$F"; [ $1 = beam ] && G="$G --workload beam"
[ $2 != 0 ] && G="$G --jitter $2"
for wto in 1 2 3; do
PFEM_AMG_W_TO=$wto timeout 900 python bench.py $G --cycle w 2>$OUT/wc2.err | tail -1 > $OUT/wc2.json
python3 -c "
import json; d=json.load(open('$OUT/wc2.json')); p=d['preconditioner']
print('$1 jitter $2 cycle', p['cycle'], p['last_level_visited_twice'], 'its', d['iterations'], 'warm', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2))"
done
done
