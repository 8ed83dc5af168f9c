#!/bin/bash
# the beam with its nodes moved off the lattice: what the smoother's knobs buy (V-cycle)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 2 --warmup 1 --no-cpu-baseline --no-jacobi-step --no-parity-step --workload beam --jitter 0.2 --cycle v"
run() {
  env "$@" timeout 900 python bench.py $F 2>$OUT/jk.err | tail -1 > $OUT/jk.json
  python3 -c "
import json; d=json.load(open('$OUT/jk.json')); p=d['preconditioner']
print('$*', 'its', d['iterations'], 'warm', round(d['ms_per_step'],2))"
}
run X=1
run PFEM_AMG_FINE_DEGREE=2
run PFEM_AMG_CHEB_DEGREE=3
run PFEM_AMG_CHEB_DEGREE=3 PFEM_AMG_FINE_DEGREE=2
run PFEM_AMG_CHEB_DEGREE=4 PFEM_AMG_FINE_DEGREE=2
run PFEM_AMG_COARSE_SCALE=1.8
run PFEM_AMG_COARSE_SCALE=1.2
run PFEM_AMG_EIG_RATIO=8
run PFEM_AMG_EIG_RATIO=30
run PFEM_AMG_PASSES=2
run PFEM_AMG_ROUNDS=32
run PFEM_AMG_NO_STRENGTH=1
