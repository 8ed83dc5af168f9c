#!/bin/bash
# bricks in one step on the hierarchy across ranks: parity cases (ranks sharing the GPU), then the self-peer probe's symbolic phases
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
if [ "${1:-all}" != "probe" ]; then
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -k "gamg" -x -q 2>&1 | tail -15 > $OUT/cb_tests_distributed.txt
tail -5 $OUT/cb_tests_distributed.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k "gamg or amg or brick" -x -q 2>&1 | tail -8 > $OUT/cb_tests_parity.txt
tail -3 $OUT/cb_tests_parity.txt
fi
show() { python3 - <<PY
import json
d=json.load(open("$1"))
for k,r in d.items():
    if isinstance(r, dict) and "ms_per_iteration" in r:
        print(f"{k:40s}", {q:(round(r[q],3) if isinstance(r[q],float) else r[q]) for q in ("iterations","ms_per_iteration","symbolic_setup_ms","numeric_setup_ms","distributed_levels","host_enqueue_ms_per_iteration") if q in r})
PY
}
PFEM_AMG_VERBOSE=1 timeout 600 python tools/probe_coupled.py 200 10 2>$OUT/cb_probe_verbose.err | grep "^{" | tail -1 > $OUT/cb_probe_verbose.json
show $OUT/cb_probe_verbose.json
timeout 600 python tools/probe_coupled.py 200 30 2>$OUT/cb_probe.err | grep "^{" | tail -1 > $OUT/cb_probe.json
show $OUT/cb_probe.json
