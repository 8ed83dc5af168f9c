#!/bin/bash
# round 6, second lease: node bricks across ranks + value codes on coupled levels: parity cases (ranks sharing the GPU), the oracle's own
# node bricks, the self-peer probe (coupled cycle over RCCL), the beam on 8 ranks sharing the GPU
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k "node_bricks or bound_and_diagonal" -x -q 2>&1 | tail -25 ) > $OUT/q2_parity.txt 2>&1
tail -4 $OUT/q2_parity.txt
( timeout 1800 python -m pytest tests/test_distributed.py -m gpu -k "gamg" -x -q 2>&1 | tail -40 ) > $OUT/q2_distributed.txt 2>&1
tail -6 $OUT/q2_distributed.txt
show() { python3 - <<PY
import json
d=json.load(open("$1"))
for k,r in d.items():
    if isinstance(r, dict) and "ms_per_iteration" in r:
        print(f"{k:40s}", {q:(round(r[q],3) if isinstance(r[q],float) else r[q]) for q in ("iterations","ms_per_iteration","symbolic_setup_ms","numeric_setup_ms","distributed_levels","host_enqueue_ms_per_iteration","rows_per_level") if q in r})
PY
}
timeout 600 python tools/probe_coupled.py 200 30 2>$OUT/q2_probe.err | grep "^{" | tail -1 > $OUT/q2_probe.json
show $OUT/q2_probe.json
timeout 900 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step --no-parity-step 2>$OUT/q2_b8.err | tail -1 > $OUT/q2_b8.json
python3 -c "
import json; d=json.load(open('$OUT/q2_b8.json')); p=d['preconditioner']
print('beam on 8 ranks sharing the GPU: its', d['iterations'], 'reason', d['converged_reason'], 'ms', round(d['ms_per_step'],1), 'first', d['first_step_ms_including_once_per_pattern_setup'], 'rows', p['rows_per_level'], 'sym', p['symbolic_setup_ms_once_per_pattern'], 'tip', d.get('check') or d.get('tip_displacement'))" || tail -5 $OUT/q2_b8.err
timeout 900 python bench.py --workload beam --steps 5 --warmup 2 --no-jacobi-step --no-pmc 2>$OUT/q2_beam1.err | tail -1 > $OUT/q2_beam1.json
python3 -c "
import json; d=json.load(open('$OUT/q2_beam1.json')); p=d['preconditioner']
print('beam on 1 rank: its', d['iterations'], 'ms', round(d['ms_per_step'],2), 'first', d['first_step_ms_including_once_per_pattern_setup'], 'rows', p['rows_per_level'])" || tail -5 $OUT/q2_beam1.err
