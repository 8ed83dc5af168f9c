#!/bin/bash
# batch size of the multigrid CG loop sized from the contraction seen so far (one rank)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -3 )
for c in 0 4 6; do
  if [ $c = 0 ]; then unset PFEM_CG_CHUNK; else export PFEM_CG_CHUNK=$c; fi
  timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('chunk $c', d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2))
"
done
unset PFEM_CG_CHUNK
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-jacobi-step --no-parity-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('cfg3', d['iterations'], round(d['ms_per_step'],2))"
