#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out
for N in 4 8; do
( timeout 1200 python bench.py --gpus $N --same-device --backend gloo --steps 1 --warmup 1 --no-strong-block --no-jacobi-step 2>$OUT/r03bg_$N.err | tail -1 ) | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print($N, d['iterations'], d['converged_reason'], d['max_nodal_error'], d['preconditioner']['rows_per_level'], d['preconditioner']['communication_per_cycle'], d['preconditioner']['levels_paired_on_the_lattice'])"
done
( timeout 900 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 0 --no-jacobi-step --no-parity-step 2>$OUT/r03bg_beam.err | tail -1 ) | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam 8', d['iterations'], d['preconditioner']['rows_per_level'])"
