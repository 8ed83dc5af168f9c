#!/bin/bash
# the block pool with splitting: parity under PFEM_DEBUG_POISON, the multi-rank cases, hipMalloc calls per phase, cold steps
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
PFEM_DEBUG_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -x -q -k "gamg or peer" 2>&1 | tail -2
F="--steps 3 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step"
for cfg in "" "--workload beam" "--cells 400 --steps 2 --warmup 1"; do
for split in 1 0; do
PFEM_POOL_SPLIT=$split PFEM_POOL_VERBOSE=1 timeout 900 python bench.py $F $cfg 2>$OUT/ps.err | tail -1 > $OUT/ps.json
python3 -c "
import json; d=json.load(open('$OUT/ps.json'))
print('[$cfg] split=$split', 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'symbolic', round(d['preconditioner']['symbolic_setup_ms_once_per_pattern'],2), 'warm', round(d['ms_per_step'],2), d['iterations'], 'mem', d['device_memory_gb'], {k:round(v,3) for k,v in d['setup_breakdown_s'].items() if 'pattern' in k})"
grep -E "gamg symbolic phase" $OUT/ps.err | tail -1; grep -E "pool:" $OUT/ps.err | sed -n 3p | cut -c1-250
done; done
