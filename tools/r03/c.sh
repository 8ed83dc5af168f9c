#!/bin/bash
# round 3, visit C: gamg knobs at 200^3 and on the beam, kernel stats of a gamg solve
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
: > $OUT/r03c_knobs.log
for cfg in "2 8 1.0" "2 8 1.5" "3 8 1.5" "2 4 1.5" "2 16 1.5" "3 16 1.5" "1 4 1.5" "2 8 1.8"; do
  set -- $cfg
  echo "== degree $1 ratio $2 scale $3" >> $OUT/r03c_knobs.log
  PFEM_AMG_CHEB_DEGREE=$1 PFEM_AMG_EIG_RATIO=$2 PFEM_AMG_COARSE_SCALE=$3 timeout 300 python tools/probe_amg.py 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'jacobi', d['jacobi']['its'], round(d['jacobi']['solve_ms'],1), 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2))
" >> $OUT/r03c_knobs.log
done
rm -rf /tmp/prof_amg
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_amg -- python3 tools/probe_amg.py 200 > $OUT/r03c_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_amg > $OUT/r03c_rocprof_kernel_stats_gamg_200.txt 2>&1
cat $OUT/r03c_knobs.log; head -40 $OUT/r03c_rocprof_kernel_stats_gamg_200.txt
