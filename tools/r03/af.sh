#!/bin/bash
# coarse-grid correction scale on 3-dof problems: config 4 beam and the small test beams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sc in 1.5 1.8 2.0; do
  export PFEM_AMG_COARSE_SCALE=$sc
  timeout 600 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('beam scale $sc', d['iterations'], round(d['ms_per_step'],1))"
  timeout 300 python tools/probe_amg.py beam:3 beam:5 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    print(' ', d['case'], 'scale $sc gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],2), 'jacobi', d.get('jacobi',{}).get('its'))
"
done
