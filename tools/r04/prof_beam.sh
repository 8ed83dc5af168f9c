#!/bin/bash
# kernel trace of the config-4 beam bench (default gamg with rigid-body modes)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_beam
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_beam -- python3 bench.py --workload beam --steps 3 --warmup 1 > $OUT/prof_beam.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_beam 45 > $OUT/prof_beam_kernel_stats.txt 2>&1
cat $OUT/prof_beam_kernel_stats.txt
tail -1 $OUT/prof_beam.log | cut -c1-600
