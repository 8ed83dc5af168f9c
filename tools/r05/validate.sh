#!/bin/bash
# after a change: the multi-rank cases (ranks sharing the GPU: gloo hooks + peer-memory transport), the parity files, one default bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_distributed.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/val_distributed.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/val_parity.txt
timeout 900 python bench.py --steps 5 --warmup 3 2>$OUT/val_bench.err | tail -1 > $OUT/val_bench.json
python3 -c "
import json; d=json.load(open('$OUT/val_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','iterations','assembly_ms_per_step','cold_value','first_step_ms_including_once_per_pattern_setup')})
print('roofline', d['roofline']['frac'], 'jacobi', (d.get('jacobi_step') or {}).get('ms_per_step'))
print('assembly_kernel', d.get('assembly_kernel'))"
