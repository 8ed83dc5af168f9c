#!/bin/bash
# is the slow first step of the first bench process on a box a matter of waiting?  (allocations on memory still being wiped)
# + config 5 on 8 ranks sharing the GPU after closing up the replicated level's positions; + the multi-rank gamg cases
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
show() { python3 -c "
import json,sys; d=json.load(open('$1'))
print('$2', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],1), 'warm', round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['setup_breakdown_s'].items()})"; }
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
timeout 600 python bench.py $F 2>/dev/null | tail -1 > $OUT/cp_1.json; show $OUT/cp_1.json "first process on the box"
timeout 600 python bench.py $F 2>/dev/null | tail -1 > $OUT/cp_2.json; show $OUT/cp_2.json "second, right behind"
sleep 20
timeout 600 python bench.py $F 2>/dev/null | tail -1 > $OUT/cp_3.json; show $OUT/cp_3.json "third, after 20 s of nothing"
timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "peer or gamg" 2>&1 | tail -2
timeout 600 python bench.py $F 2>/dev/null | tail -1 > $OUT/cp_4.json; show $OUT/cp_4.json "fourth, right behind a pytest of 39 multi-process cases"
PFEM_POOL_VERBOSE=1 timeout 600 python bench.py $F 2>$OUT/cp_5.err | tail -1 > $OUT/cp_5.json; show $OUT/cp_5.json "fifth"; grep "pool:" $OUT/cp_5.err | head -8
( timeout 1500 python bench.py --gpus 8 --same-device --backend gloo --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step 2>$OUT/cp_8ranks.err | tail -1 ) > $OUT/cp_cfg5_8ranks.json
python3 -c "
import json; c=json.load(open('$OUT/cp_cfg5_8ranks.json')); p=c['preconditioner']
print('cfg5 x8', c['iterations'], p['rows_per_level'], p.get('distributed_levels'), c['ms_per_step'])"
