#!/bin/bash
# kernel timeline of the cold first step at config 3 (tools/trace_first_step.py): where the time between the kernels goes
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_first
timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/prof_first -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/first_step_prof.log 2>&1
python tools/trace_first_step.py /tmp/prof_first 1 > $OUT/first_step_timeline.txt 2>&1
python tools/trace_first_step.py /tmp/prof_first 2 | head -3 >> $OUT/first_step_timeline.txt 2>&1
cat $OUT/first_step_timeline.txt | cut -c1-200
tail -1 $OUT/first_step_prof.log | cut -c1-300
