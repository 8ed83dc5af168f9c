#!/bin/bash
# kernel trace of the default bench (per-kernel averages, the iteration's timeline) + A/B lines on the same box
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
TAG=${1:-p1}
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/${TAG}_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats 45 > $OUT/${TAG}_kernel_stats.txt 2>&1
python tools/trace_gaps.py /tmp/prof_stats k_pc_update > $OUT/${TAG}_timeline.txt 2>&1
python tools/trace_phase.py /tmp/prof_stats > $OUT/${TAG}_phase.txt 2>&1
head -60 $OUT/${TAG}_timeline.txt
grep -E "k_spmvr_vd|k_amg_tail|k_gather|k_amg_max_rows|k_amg_diag_bound|k_lat_galerkin|k_vd_encode" $OUT/${TAG}_kernel_stats.txt
for v in "" "PFEM_AMG_TAIL_LDS=0"; do
  ( env $v timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-jacobi-step 2>/dev/null | tail -1 ) > $OUT/${TAG}_ab.json
  python3 -c "
import json; d=json.load(open('$OUT/${TAG}_ab.json')); print('bench [$v]', round(d['ms_per_step'],3), d['iterations'], round(d['ms_per_iteration'],4), round(d['roofline']['avg_launch_ms'],4), round(d['assembly_ms_per_step'],3), round(d['preconditioner']['numeric_setup_ms_per_solve_inside_the_timer'],3))"
done
