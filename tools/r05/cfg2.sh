#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
for vd in 1 0; do
for fmt in auto; do
PFEM_SPMV_VALDICT=$vd timeout 900 python bench.py --cells 100 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/c2.json
python3 -c "
import json; d=json.load(open('$OUT/c2.json')); r=d['roofline']
print('cfg2 valdict=$vd its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'jacobi', round(d['jacobi_step']['ms_per_step'],2), 'spmv ms', round(r['avg_launch_ms'],4), r['kernel'][:40])"
done
done
timeout 900 python bench.py --workload beam --cells 25 --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step 2>/dev/null | tail -1 > $OUT/c2b.json
python3 -c "
import json; d=json.load(open('$OUT/c2b.json')); r=d['roofline']
print('small beam', d['config']['free_dofs'], 'its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'spmv ms', round(r['avg_launch_ms'],4), r['kernel'][:40])"
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -3
