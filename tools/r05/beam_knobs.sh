#!/bin/bash
# config 4 with its level 0 in bricks of 4: the over-correction and the smoothing interval once more
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam"
run() {
  env "$@" timeout 900 python bench.py $F 2>/dev/null | tail -1 > $OUT/bk.json
  python3 -c "
import json; d=json.load(open('$OUT/bk.json'))
print('$*', 'its', d['iterations'], 'warm', round(d['ms_per_step'],2))"
}
run X=1
run PFEM_AMG_COARSE_SCALE=1.2
run PFEM_AMG_COARSE_SCALE=1.8
run PFEM_AMG_COARSE_SCALE=2.0
run PFEM_AMG_EIG_RATIO=8
run PFEM_AMG_EIG_RATIO=30
run PFEM_AMG_FINE_DEGREE=2
run PFEM_AMG_COARSE_SCALE=1.8 PFEM_AMG_EIG_RATIO=30
