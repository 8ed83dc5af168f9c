#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( PFEM_AMG_VERBOSE=1 timeout 600 python -m pytest tests/test_distributed.py -m gpu -k "elast-3-yslabs-gamg and ranks_on_one" -x -q -s 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -120 ) > $OUT/dbg1.txt 2>&1
grep -n "gamg symbolic\|refused\|passed\|failed" $OUT/dbg1.txt | head -60
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k "node_bricks or fused_cycle or restatement" -x -q 2>&1 | tail -15 ) > $OUT/dbg1_parity.txt 2>&1
tail -5 $OUT/dbg1_parity.txt
( timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-jacobi-step 2>$OUT/dbg1_bench.err | tail -1 ) > $OUT/dbg1_bench.json
python3 -c "
import json; d=json.load(open('$OUT/dbg1_bench.json')); print('bench', d['ms_per_step'], d['iterations'], d['ms_per_iteration'], d['roofline']['avg_launch_ms'])"
