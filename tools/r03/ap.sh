#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -3 )
timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2), 'oracle', d['gamg'].get('oracle',{}).get('its'))
"
