#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r03g_knobs.log
for cfg in "2 1 8 1.5 1" "2 1 8 1.5 0" "2 1 8 1.8 1" "3 1 8 1.8 1" "2 1 8 2.0 1" "2 1 16 1.8 1"; do
  set -- $cfg
  echo "== degree $1 fine_degree $2 ratio $3 scale $4 graph $5" >> $OUT/r03g_knobs.log
  PFEM_CG_GRAPH=$5 PFEM_AMG_CHEB_DEGREE=$1 PFEM_AMG_FINE_DEGREE=$2 PFEM_AMG_EIG_RATIO=$3 PFEM_AMG_COARSE_SCALE=$4 timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'jacobi', d['jacobi']['its'], round(d['jacobi']['solve_ms'],1), 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2))
" >> $OUT/r03g_knobs.log
done
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -5 ) > $OUT/r03g_gamg_tests.log 2>&1
cat $OUT/r03g_knobs.log; tail -3 $OUT/r03g_gamg_tests.log
