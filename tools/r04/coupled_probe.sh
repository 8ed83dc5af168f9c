#!/bin/bash
# self-peer probe of the coupled cycle over RCCL: replication threshold sweep, the cycle as a graph, kernel trace
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for G in 0 1; do
  PFEM_AMG_COUPLED_GRAPH=$G timeout 900 python tools/probe_coupled.py 200 30 2>$OUT/probe_coupled_graph$G.err | grep "^{" | tail -1 > $OUT/probe_coupled_graph$G.json
  python3 - <<PY
import json
d=json.load(open("$OUT/probe_coupled_graph$G.json"))
for k,r in d.items():
    if isinstance(r, dict) and "ms_per_iteration" in r:
        print("graph=$G", f"{k:40s}", {q:(round(r[q],4) if isinstance(r[q],float) else r[q]) for q in ("ms_per_iteration","host_enqueue_ms_per_iteration","distributed_levels","exchanges_per_cycle","allreduces_per_cycle") if q in r})
PY
done
rm -rf /tmp/prof_cp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_cp -- python3 tools/probe_coupled.py 200 30 > $OUT/prof_cp.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_cp 40 > $OUT/probe_coupled_kernel_stats.txt 2>&1
head -45 $OUT/probe_coupled_kernel_stats.txt
