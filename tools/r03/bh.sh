#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in 0 3 4 6 12 13; do
  if [ $c = 0 ]; then unset PFEM_CG_CHUNK; else export PFEM_CG_CHUNK=$c; fi
  timeout 300 python tools/probe_amg.py 100 200 2>&1 | python3 -c "
import sys, json
out=[]
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    out.append('%s: %d its %.2f ms' % (d['case'], d['gamg']['its'], d['gamg']['solve_ms']))
print('chunk $c', ' | '.join(out))
"
done
