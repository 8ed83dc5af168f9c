#!/bin/bash
# one kernel trace per loop + the default bench line with both event-pair figures
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/final_bench_n1.err | tail -1 ) > $OUT/final_bench_n1.json
rm -rf /tmp/prof_stats /tmp/prof_stats_j
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/final_prof_stats.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats > $OUT/final_rocprofv3_kernel_stats.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_j -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --pc jacobi > $OUT/final_prof_stats_j.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_j > $OUT/final_rocprofv3_kernel_stats_jacobi_loop.txt 2>&1
python3 -c "
import json
d=json.load(open('$OUT/final_bench_n1.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['avg_launch_ms_in_jacobi_step'], d['jacobi_step']['ms_per_step'])"
head -8 $OUT/final_rocprofv3_kernel_stats.txt; head -6 $OUT/final_rocprofv3_kernel_stats_jacobi_loop.txt
