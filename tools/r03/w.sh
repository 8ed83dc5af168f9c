#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg or renumbering" 2>&1 | tail -6 )
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -3 )
for f in 0 1; do echo "== rap_sort $f"; if [ $f = 1 ]; then export PFEM_AMG_RAP_SORT=1; else unset PFEM_AMG_RAP_SORT; fi; timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym_ms', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['nnz'][:3])
"; done
unset PFEM_AMG_RAP_SORT
( timeout 900 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 0 --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam 8 ranks', d['iterations'], round(d['ms_per_step'],1), d['preconditioner'].get('rows_per_level'))" )
