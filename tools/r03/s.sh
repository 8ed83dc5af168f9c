#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or renumbering" 2>&1 | tail -4 )
for i in 1 2; do
PFEM_AMG_VERBOSE=1 timeout 600 python bench.py --cells 400 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2> gpurun_out/r03s_cfg5_verbose.err | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('cfg5', d['iterations'], round(d['ms_per_step'],1), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'sym', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],1))"
grep "gamg symbolic level 0 " gpurun_out/r03s_cfg5_verbose.err | head -12
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('cfg3', d['iterations'], round(d['ms_per_step'],1), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'sym', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],1), d['setup_breakdown_s'])"
