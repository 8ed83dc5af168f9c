#!/bin/bash
# value dictionary of the SpMV: parity, then configs 3 and 4 with and without it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step"
for wl in poisson beam; do
G="$F"; [ $wl = beam ] && G="$G --workload beam"
for vd in 0 1; do
PFEM_SPMV_VALDICT=$vd PFEM_VD_VERBOSE=1 timeout 900 python bench.py $G 2>$OUT/vd_${wl}_$vd.err | tail -1 > $OUT/vd_${wl}_$vd.json
python3 -c "
import json; d=json.load(open('$OUT/vd_${wl}_$vd.json')); r=d['roofline']
print('$wl valdict=$vd its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'jacobi', d.get('jacobi_step',{}).get('ms_per_step'), 'spmv ms', r.get('avg_launch_ms'), 'rnorm', d['rnorm'])"
grep "value dictionary" $OUT/vd_${wl}_$vd.err | sort | uniq -c | sort -rn | head -3
done
done
