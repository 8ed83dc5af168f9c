#!/bin/bash
# W-cycle on matched aggregates: parity cases, then the four benches (lattice / moved nodes x cube / beam) under V and W
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -15
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
for wl in "poisson 0.2" "beam 0.2" "poisson 0" "beam 0"; do
set -- $wl
G="$F"; [ $1 = beam ] && G="$G --workload beam"
[ $2 != 0 ] && G="$G --jitter $2"
for cyc in v w; do
timeout 900 python bench.py $G --cycle $cyc 2>$OUT/wc_$1_$2_$cyc.err | tail -1 > $OUT/wc_$1_$2_$cyc.json
python3 -c "
import json; d=json.load(open('$OUT/wc_$1_$2_$cyc.json')); p=d['preconditioner']
print('$1 jitter $2 cycle', p['cycle'], p['last_level_visited_twice'], 'its', d['iterations'], 'warm', round(d['ms_per_step'],2), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'rows', p['rows_per_level'], 'rnorm', d['rnorm'])"
done
done
