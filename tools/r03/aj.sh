#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$OUT/aj.err | tail -1 ) > $OUT/aj_bench_n1.json
python3 -c "
import json
d=json.load(open('$OUT/aj_bench_n1.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['avg_launch_ms_in_jacobi_step'], d['jacobi_step']['ms_per_step'], r['event_pair_offset_ms_not_subtracted'])"
