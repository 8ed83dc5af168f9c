#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden_drivers.py tests/test_gpu_full_size.py -m gpu -x -q -k "gamg" 2>&1 | tail -6 )
( timeout 1200 python -m pytest tests/test_distributed.py -m gpu -q -s -k "gamg or rccl" 2>&1 | grep -E "^gamg elast|passed|failed|Error|assert" | tail -20 )
timeout 600 python bench.py --workload beam --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam', d['iterations'], round(d['ms_per_step'],1), d['preconditioner']['coarse_scale'], d['max_displacement_magnitude_owned_rows'] if 'max_displacement_magnitude_owned_rows' in d else '')"
