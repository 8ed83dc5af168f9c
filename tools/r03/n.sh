#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "renumbering or gamg" 2>&1 | tail -30 ) > $OUT/r03n_tests.log 2>&1
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "reorder or gamg" 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -60 ) > $OUT/r03n_dist.log 2>&1
: > $OUT/r03n_knobs.log
for nh in 0 1; do
  echo "== no_hint $nh" >> $OUT/r03n_knobs.log
  if [ $nh = 1 ]; then export PFEM_AMG_NO_HINT=1; else unset PFEM_AMG_NO_HINT; fi
  timeout 300 python tools/probe_amg.py 60 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'jacobi', d['jacobi']['its'], round(d['jacobi']['solve_ms'],1), 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'rows', d['gamg']['hierarchy']['rows'], 'sym_ms', round(d['gamg']['hierarchy']['symbolic_ms'],1), 'oracle', d['gamg'].get('oracle',{}).get('its'))
" >> $OUT/r03n_knobs.log
done
tail -8 $OUT/r03n_tests.log; tail -40 $OUT/r03n_dist.log; cat $OUT/r03n_knobs.log
