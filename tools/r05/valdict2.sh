#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
F="--steps 3 --warmup 2 --no-cpu-baseline --no-parity-step --no-jacobi-step"
for wl in poisson beam; do
G="$F"; [ $wl = beam ] && G="$G --workload beam"
rm -rf /tmp/prof_vd
PFEM_VD_VERBOSE=1 timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_vd -- python3 bench.py $G 2>$OUT/vd2_$wl.err | tail -1 > $OUT/vd2_$wl.json
python3 -c "
import json; d=json.load(open('$OUT/vd2_$wl.json')); r=d['roofline']
print('$wl its', d['iterations'], 'warm', round(d['ms_per_step'],3), 'first', round(d['first_step_ms_including_once_per_pattern_setup'],2), 'spmv ms', r.get('avg_launch_ms'))"
python tools/summarize_prof.py stats /tmp/prof_vd 16 2>&1 | tee $OUT/vd2_${wl}_stats.txt | head -22
done
