#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam"
for split in 0 1 0 1; do
PFEM_POOL_SPLIT=$split PFEM_POOL_VERBOSE=1 PFEM_AMG_VERBOSE=1 timeout 900 python bench.py $F 2>$OUT/psb_$split.err | tail -1 > $OUT/psb.json
python3 -c "
import json; d=json.load(open('$OUT/psb.json'))
print('beam split=$split', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2))"
grep -E "gamg symbolic phase" $OUT/psb_$split.err | tail -1
done
awk '/nodes of level 0/{c++} c==2' $OUT/psb_1.err | grep "symbolic level" | sort -t' ' -k8 -n -r | head -8
