#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/phase_prof.log 2>&1
python tools/trace_phase.py /tmp/prof_stats > $OUT/final_kernel_phases_of_a_step.txt 2>&1
head -8 $OUT/final_kernel_phases_of_a_step.txt; tail -2 $OUT/final_kernel_phases_of_a_step.txt
