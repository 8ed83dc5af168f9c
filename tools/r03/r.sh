#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
: > $OUT/r03r_knobs.log
run() { echo "== $1" >> $OUT/r03r_knobs.log; timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'sym_ms', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['rows'])
" >> $OUT/r03r_knobs.log; }
run default
PFEM_AMG_ROUNDS=3 run "rounds 3"
PFEM_AMG_ROUNDS=2 run "rounds 2"
PFEM_AMG_NO_STRENGTH=1 run "no strength"
rm -rf /tmp/prof_sym
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_sym -- python3 tools/probe_amg.py 200 > $OUT/r03r_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_sym > $OUT/r03r_rocprof_probe200.txt 2>&1
cat $OUT/r03r_knobs.log; head -40 $OUT/r03r_rocprof_probe200.txt
