#!/bin/bash
# work buffers of the gamg symbolic phase reused across levels: parity + phase timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or renumbering" 2>&1 | tail -5 )
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -3 )
timeout 300 python tools/probe_amg.py 60 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym_ms', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['rows'], 'oracle', d['gamg'].get('oracle',{}).get('its'))
"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('cfg3', d['iterations'], round(d['ms_per_step'],1), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],1), 'sym', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],1))"
