#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python tools/probe_amg.py 60 100 160 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:200].rstrip()); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), d['gamg']['hierarchy']['rows'])
"
( timeout 1200 python -m pytest tests/test_distributed.py tests/test_gpu_parity.py -m gpu -q -k "gamg or rccl" 2>&1 | tail -3 )
bash tools/r03/ab.sh 2>&1 | tail -7 | head -4
