#!/bin/bash
# lattice pairing as the default: parity (one rank and several), then the sizes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_golden_drivers.py -m gpu -q -k "gamg" 2>&1 | tail -25 )
( timeout 1200 python -m pytest tests/test_distributed.py -m gpu -q -s -k "gamg or rccl" 2>&1 | grep -E "^gamg|passed|failed|Error|assert|^E " | tail -40 )
timeout 900 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:200].rstrip()); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['rows'])
"
