#!/bin/bash
# every kernel of the default (multigrid) bench, not only the top 25: the small ones of the numeric phase and the cycle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=$GRAFT_REPO_ROOT/gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/prof_all
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_all -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/r03ar.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_all 90 > $OUT/r03ar_kernel_stats_all.txt 2>&1
grep -v "rocprim\|k_amg_match\|k_build\|k_inc\|k_amg_emit\|k_amg_run\|k_amg_agg\|k_amg_compose\|k_amg_hint\|k_amg_graph\|k_row\|k_fill\|k_slice\|k_sell\|k_amg_mark\|k_amg_assign\|k_amg_count\|k_amg_iota\|k_amg_dst\|k_morton\|k_box\|k_pack_node\|k_perm" $OUT/r03ar_kernel_stats_all.txt | cut -c1-135
