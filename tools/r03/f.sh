#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
: > $OUT/r03f_knobs.log
for cfg in "2 0 8 1.5" "2 1 8 1.5" "3 1 8 1.5" "3 1 4 1.5" "4 1 8 1.5" "3 1 8 1.8" "3 2 8 1.5"; do
  set -- $cfg
  echo "== degree $1 fine_degree $2 ratio $3 scale $4" >> $OUT/r03f_knobs.log
  PFEM_AMG_CHEB_DEGREE=$1 PFEM_AMG_FINE_DEGREE=$2 PFEM_AMG_EIG_RATIO=$3 PFEM_AMG_COARSE_SCALE=$4 timeout 300 python tools/probe_amg.py 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'jacobi', d['jacobi']['its'], round(d['jacobi']['solve_ms'],1), 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2))
" >> $OUT/r03f_knobs.log
done
( timeout 900 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg" 2>&1 | tail -5 ) > $OUT/r03f_dist_gamg.log 2>&1
rm -rf /tmp/prof_amg
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_amg -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/r03f_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_amg > $OUT/r03f_rocprof_kernel_stats_gamg_bench.txt 2>&1
cat $OUT/r03f_knobs.log; tail -3 $OUT/r03f_dist_gamg.log; head -32 $OUT/r03f_rocprof_kernel_stats_gamg_bench.txt
