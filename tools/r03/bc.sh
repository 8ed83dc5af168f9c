#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_oracle.py -m gpu -q -k "gamg or rccl" 2>&1 | tail -3 )
for cfg in "" "PFEM_AMG_EIG_RATIO=8"; do
env $cfg timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('[$cfg] cfg5', d['iterations'], round(d['ms_per_step'],1), d['max_nodal_error'], d['preconditioner']['eig_ratio'])"
done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('cfg3', d['iterations'], round(d['ms_per_step'],2), d['max_nodal_error'], d['parity_tolerance_step']['iterations'], d['parity_tolerance_step']['max_nodal_error'])"
timeout 900 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam', d['iterations'], round(d['ms_per_step'],2), d['preconditioner']['eig_ratio'], d['preconditioner']['coarse_scale'])"
