#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "" "PFEM_AMG_PASSES0=4" "PFEM_AMG_PASSES=4" "PFEM_AMG_PASSES0=2" "PFEM_AMG_PASSES0=6"; do
  env $cfg timeout 600 python tools/probe_amg.py 100 160 200 2>&1 | python3 -c "
import sys, json
out=[]
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    out.append('%s: %d its %.2f ms' % (d['case'], d['gamg']['its'], d['gamg']['solve_ms']))
print('[$cfg]', ' | '.join(out))
"
done
