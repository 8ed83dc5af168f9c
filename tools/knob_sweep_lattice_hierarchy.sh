#!/bin/bash
# knobs again, now that the hierarchy is the lattice's: coarse-correction scale, smoothing interval, degrees (config 3, 160^3 and 400^3/cfg5 spot checks)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03bb_knobs.txt
for cfg in "" "PFEM_AMG_COARSE_SCALE=1.3" "PFEM_AMG_COARSE_SCALE=1.7" "PFEM_AMG_COARSE_SCALE=2.0" "PFEM_AMG_EIG_RATIO=4" "PFEM_AMG_EIG_RATIO=16" "PFEM_AMG_CHEB_DEGREE=3" "PFEM_AMG_CHEB_DEGREE=1" "PFEM_AMG_FINE_DEGREE=2" "PFEM_AMG_COARSE_SCALE=1.7 PFEM_AMG_EIG_RATIO=16" "PFEM_AMG_COARSE_SCALE=2.0 PFEM_AMG_EIG_RATIO=16"; do
  env $cfg timeout 600 python tools/probe_amg.py 100 160 200 2>&1 | python3 -c "
import sys, json
out=[]
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    out.append('%s: %d its %.2f ms' % (d['case'], d['gamg']['its'], d['gamg']['solve_ms']))
print('[$cfg]', ' | '.join(out))
" >> gpurun_out/r03bb_knobs.txt
done
cat gpurun_out/r03bb_knobs.txt
