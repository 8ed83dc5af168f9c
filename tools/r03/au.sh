#!/bin/bash
# 16-node aggregates on level 0 against 8: where is the crossover
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
: > gpurun_out/r03au_passes0.txt
for cfg in "" "PFEM_AMG_PASSES0=4"; do
  env $cfg timeout 900 python tools/probe_amg.py 128 160 200 256 320 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print('[$cfg]', d['case'], 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'numeric', round(d['gamg']['hierarchy']['numeric_ms'],2), d['gamg']['hierarchy']['rows'][:3])
" >> gpurun_out/r03au_passes0.txt
done
for cfg in "" "PFEM_AMG_PASSES0=4"; do
  env $cfg timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('[$cfg] cfg5', d['iterations'], round(d['ms_per_step'],1), d['max_nodal_error'])" >> gpurun_out/r03au_passes0.txt
done
cat gpurun_out/r03au_passes0.txt
