#!/bin/bash
# compat path at config 2, gamg without a lattice (--jitter), the N>1 line's new fields with 2 ranks sharing the GPU
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "poisson or assembl or gather or tet10" 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-jacobi-step --no-parity-step 2>/dev/null | tail -1 > $OUT/ma_bench.json
python3 -c "
import json; d=json.load(open('$OUT/ma_bench.json'))
print('default', {k:d[k] for k in ('value','ms_per_step','iterations','assembly_ms_per_step','first_step_ms_including_once_per_pattern_setup')})"
for pc in jacobi gamg; do
  timeout 1200 python bench.py --mode compat --cells 100 --pc $pc --steps 2 > $OUT/ma_compat_$pc.json 2> $OUT/ma_compat_$pc.err
  python3 -c "
import json; d=json.load(open('$OUT/ma_compat_$pc.json'))
print('compat $pc', {k:(round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ('oracle_serial_assembly_s','insert_values_pass_s','set_zero_pattern_on_device_s','element_loop_add_values_s','factorise_and_solve_s','dof_per_s_reference_timed_region','element_loop_vs_oracle_serial_assembly','solver_line')})" || tail -5 $OUT/ma_compat_$pc.err
done
timeout 900 python bench.py --jitter 0.2 --steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step 2>$OUT/ma_jitter.err | tail -1 > $OUT/ma_jitter.json
python3 -c "
import json; d=json.load(open('$OUT/ma_jitter.json'))
print('jitter cube', {k:d.get(k) for k in ('value','ms_per_step','iterations','max_nodal_error','first_step_ms_including_once_per_pattern_setup')}, d['preconditioner']['rows_per_level'], d['preconditioner']['operator_complexity'], d.get('parity_tolerance_step'))" || tail -5 $OUT/ma_jitter.err
timeout 900 python bench.py --workload beam --jitter 0.2 --steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step 2>$OUT/ma_jitter_beam.err | tail -1 > $OUT/ma_jitter_beam.json
python3 -c "
import json; d=json.load(open('$OUT/ma_jitter_beam.json'))
print('jitter beam', {k:d.get(k) for k in ('value','ms_per_step','iterations','max_displacement_magnitude_owned_rows','first_step_ms_including_once_per_pattern_setup')}, d['preconditioner']['rows_per_level'], d['preconditioner']['operator_complexity'])" || tail -5 $OUT/ma_jitter_beam.err
timeout 1200 python bench.py --gpus 2 --same-device --cells 100 --steps 3 --warmup 2 2>$OUT/ma_n2.err | tail -1 > $OUT/ma_n2.json
python3 -c "
import json; d=json.load(open('$OUT/ma_n2.json'))
print('n2', {k:d.get(k) for k in ('value','ms_per_step','iterations','cold_value')}, 'jacobi', (d.get('jacobi_step') or {}).get('ms_per_step'))
c=d['comm']; print(c.get('transport'), c.get('link_latencies')); print(c.get('coupled_cycle')); print(json.dumps(c.get('transports'))[:1500])" || tail -20 $OUT/ma_n2.err
