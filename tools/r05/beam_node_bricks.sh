#!/bin/bash
# config 4 (beam): node bricks in one step -- parity cases with rigid-body modes, the full-size beam, bench line cold / warm, symbolic phases
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -k "rbm or rigid or elast or gamg or cook" -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -k "beam" -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_distributed.py -m gpu -k "gamg and elast" -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 600 python bench.py --workload beam --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-parity-step 2>/dev/null | tail -1 > $OUT/beam_bricks_$i.json
python3 -c "
import json; d=json.load(open('$OUT/beam_bricks_$i.json'))
print({k:d[k] for k in ('ms_per_step','iterations','first_step_ms_including_once_per_pattern_setup','assembly_ms_per_step')}, d['preconditioner']['rows_per_level'], {k:v for k,v in d['preconditioner'].items() if 'symbolic' in k or 'numeric' in k})"
done
PFEM_AMG_VERBOSE=1 timeout 600 python bench.py --workload beam --steps 2 --warmup 1 --no-cpu-baseline --no-jacobi-step --no-parity-step > /dev/null 2> $OUT/beam_bricks_verbose.err
awk '/nodes of level 0/{c++} c==2' $OUT/beam_bricks_verbose.err | grep symbolic | head -60
