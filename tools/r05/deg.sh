#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step --no-jacobi-step"
for wl in beam poisson; do
G="$F"; [ $wl = beam ] && G="$G --workload beam"
for deg in 2 1 3; do
PFEM_AMG_CHEB_DEGREE=$deg timeout 900 python bench.py $G 2>/dev/null | tail -1 > $OUT/deg.json
python3 -c "
import json; d=json.load(open('$OUT/deg.json'))
print('$wl coarse degree $deg its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'ms/it', round(d['ms_per_iteration'],3))"
done
done
