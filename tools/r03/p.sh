#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
: > $OUT/r03p_knobs.log
for cfg in "2 1 8 1.5" "2 1 8 1.2" "2 1 8 1.8" "2 2 8 1.5" "3 1 8 1.5" "2 1 4 1.5" "1 1 8 1.5"; do
  set -- $cfg
  echo "== degree $1 fine_degree $2 ratio $3 scale $4" >> $OUT/r03p_knobs.log
  PFEM_AMG_CHEB_DEGREE=$1 PFEM_AMG_FINE_DEGREE=$2 PFEM_AMG_EIG_RATIO=$3 PFEM_AMG_COARSE_SCALE=$4 timeout 300 python tools/probe_amg.py 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym_ms', round(d['gamg']['hierarchy']['symbolic_ms'],1))
" >> $OUT/r03p_knobs.log
done
rm -rf /tmp/prof_amg
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_amg -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/r03p_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_amg > $OUT/r03p_rocprof_kernel_stats_gamg_bench.txt 2>&1
cat $OUT/r03p_knobs.log; head -30 $OUT/r03p_rocprof_kernel_stats_gamg_bench.txt
