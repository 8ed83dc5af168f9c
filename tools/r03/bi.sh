#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for t in 1024 2304 400 0; do
  export PFEM_AMG_TAIL_ROWS=$t
  timeout 300 python tools/probe_amg.py 100 200 2>&1 | python3 -c "
import sys, json
out=[]
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    out.append('%s: %d its %.2f ms' % (d['case'], d['gamg']['its'], d['gamg']['solve_ms']))
print('tail rows $t', ' | '.join(out))
"
done
