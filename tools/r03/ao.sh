#!/bin/bash
# kernel trace of the coupled-cycle probe (rank = its own neighbour over RCCL): what the exchanges cost on the device
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_cpl
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_cpl -- python3 tools/probe_coupled.py 200 30 > $OUT/r03ao_probe.json 2> $OUT/r03ao.err
python tools/summarize_prof.py stats /tmp/prof_cpl > $OUT/r03ao_kernel_stats_coupled_probe.txt 2>&1
head -40 $OUT/r03ao_kernel_stats_coupled_probe.txt | cut -c1-140
grep -i "nccl\|rccl\|pack_send\|unpack_sum" $OUT/r03ao_kernel_stats_coupled_probe.txt | cut -c1-160
