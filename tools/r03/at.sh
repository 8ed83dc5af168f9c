#!/bin/bash
# aggregate size: 2^passes nodes per aggregate on all levels / on level 0 only
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03at_passes.txt
for cfg in "" "PFEM_AMG_PASSES=2" "PFEM_AMG_PASSES=4" "PFEM_AMG_PASSES0=4" "PFEM_AMG_PASSES0=2" "PFEM_AMG_PASSES0=4 PFEM_AMG_COARSE_SCALE=1.8"; do
  env $cfg timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('[$cfg]', d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric', round(d['gamg']['hierarchy']['numeric_ms'],2), d['gamg']['hierarchy']['rows'][:4])
" >> gpurun_out/r03at_passes.txt
done
cat gpurun_out/r03at_passes.txt
