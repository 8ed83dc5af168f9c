#!/bin/bash
# lattice pairing: the odd node of a line joins its neighbour's pair (1) or stays alone (0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03aw_absorb.txt
for cfg in "PFEM_AMG_LATTICE_ABSORB=0" "PFEM_AMG_LATTICE_ABSORB=0 PFEM_AMG_COARSE_SCALE=1.8"; do
  env $cfg timeout 900 python tools/probe_amg.py 60 100 128 160 200 256 beam:5 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:200].rstrip()); continue
    print('[$cfg]', d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['rows'])
" >> gpurun_out/r03aw_absorb.txt
done
cat gpurun_out/r03aw_absorb.txt
PFEM_AMG_VERBOSE=1 timeout 300 python tools/probe_amg.py 200 2>&1 | grep "gamg symbolic" | head -30
