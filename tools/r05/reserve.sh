#!/bin/bash
# pool reserve at the pattern build: the beam's and the cube's first solve with and without it (PFEM_POOL_RESERVE)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for wl in beam poisson; do
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
[ $wl = beam ] && F="$F --workload beam"
for r in 0 1 0 1; do
PFEM_POOL_RESERVE=$r PFEM_POOL_VERBOSE=1 timeout 900 python bench.py $F 2>$OUT/rs_${wl}_$r.err | tail -1 > $OUT/rs.json
python3 -c "
import json; d=json.load(open('$OUT/rs.json'))
def walk(o):
    if isinstance(o,dict):
        for k,v in o.items():
            if isinstance(v,(dict,list)): yield from walk(v)
            else: yield k,v
    elif isinstance(o,list):
        for v in o: yield from walk(v)
print('$wl reserve=$r', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), d['iterations'], 'pattern_s', [round(v,3) for k,v in walk(d) if k.startswith('symbolic_pattern_and_incidence')])"
grep -E "gamg symbolic phase" $OUT/rs_${wl}_$r.err | tail -1
done
done
grep "pool:" $OUT/rs_beam_1.err | head -4
