#!/bin/bash
# the inverse diagonal as value codes in the Jacobi loop: parity at full size, then the Jacobi step with and without
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step"
for wl in poisson beam; do
G="$F"; [ $wl = beam ] && G="$G --workload beam"
timeout 900 python bench.py $G 2>/dev/null | tail -1 > $OUT/dinv.json
python3 -c "
import json; d=json.load(open('$OUT/dinv.json')); j=d['jacobi_step']
print('$wl warm', round(d['ms_per_step'],3), 'jacobi', round(j['ms_per_step'],2), j['iterations'], 'ms/it', round(j['ms_per_step']/j['iterations'],4))"
done
rm -rf /tmp/prof_dj
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_dj -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --pc jacobi > /dev/null 2>&1
python tools/summarize_prof.py stats /tmp/prof_dj 8 | cut -c1-60,95-140
