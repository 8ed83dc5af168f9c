#!/bin/bash
# peer transport after the multi-block all-reduce / exchange, compat path after the ADD-pass changes, N=2 same-device line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_distributed.py -m gpu -x -q -k "peer or gamg" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fortran_boundary.py tests/test_golden_drivers.py -m gpu -x -q 2>&1 | tail -3
for pc in jacobi; do
  timeout 1200 python bench.py --mode compat --cells 100 --pc $pc --steps 3 > $OUT/mb_compat_$pc.json 2> $OUT/mb_compat_$pc.err
  python3 -c "
import json; d=json.load(open('$OUT/mb_compat_$pc.json'))
print('compat $pc', {k:(round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ('oracle_serial_assembly_s','insert_values_pass_s','set_zero_pattern_on_device_s','element_loop_add_values_s','factorise_and_solve_s','dof_per_s_reference_timed_region','element_loop_vs_oracle_serial_assembly','solver_line')})" || tail -5 $OUT/mb_compat_$pc.err
done
timeout 1200 python bench.py --gpus 2 --same-device --cells 100 --steps 3 --warmup 2 2>$OUT/mb_n2.err | tail -1 > $OUT/mb_n2.json
python3 -c "
import json; d=json.load(open('$OUT/mb_n2.json'))
print('n2', {k:d.get(k) for k in ('value','ms_per_step','iterations','cold_value')})
t=d['comm']['transports']
for k,v in t.items():
    if isinstance(v, dict): print(k, {q:v.get(q) for q in ('value','ms_per_step','iterations','ms_per_iteration','host_enqueue_us_per_iteration','link_latencies','coupled_cycle','skipped')})" || tail -20 $OUT/mb_n2.err
python tools/probe_peer.py 2>/dev/null | tail -1 > $OUT/mb_probe_peer.json; head -c 1500 $OUT/mb_probe_peer.json
