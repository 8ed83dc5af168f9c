#!/bin/bash
# pairing on the lattice (bricks of 2, the odd node joins its line neighbour) against the curve-rank pairing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03av_lattice.txt
for cfg in "" "PFEM_AMG_NO_LATTICE=1"; do
  env $cfg timeout 900 python tools/probe_amg.py 60 100 128 160 200 256 beam:5 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:200].rstrip()); continue
    print('[$cfg]', d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric', round(d['gamg']['hierarchy']['numeric_ms'],2), 'sym', round(d['gamg']['hierarchy']['symbolic_ms'],1), d['gamg']['hierarchy']['rows'], 'oracle', d['gamg'].get('oracle',{}).get('its'))
" >> gpurun_out/r03av_lattice.txt
done
cat gpurun_out/r03av_lattice.txt
