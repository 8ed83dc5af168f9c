#!/bin/bash
# round 6, evidence at HEAD ($1 = short commit): driver-form bench lines (N=1 default with its in-run PMC passes, beam, config 5 alone, 100^3,
# fp64-values variants, 8 ranks sharing the GPU for configs 5 and 4), kernel traces (default, Jacobi loop, beam) with the iteration's
# timeline and the step's phase listing, SQ counters of the assembly kernel, the coupled cycle over RCCL (self-peer), a non-box
# partition at size, 2 ranks over both transports, moved-mesh and renumbered lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
export PFEM_HEAD=${1:-unknown}
echo "$PFEM_HEAD" > $OUT/final_head.txt
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/final_bench_n1.err | tail -1 ) > $OUT/final_bench_n1.json
( timeout 900 python bench.py --workload beam --steps 5 --warmup 2 2>$OUT/final_bench_beam.err | tail -1 ) > $OUT/final_bench_beam.json
( PFEM_SPMV_VALDICT=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/final_bench_n1_fp64_values.err | tail -1 ) > $OUT/final_bench_n1_fp64_values.json
( timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline 2>$OUT/final_bench_cfg5.err | tail -1 ) > $OUT/final_bench_cfg5_single_gpu.json
( timeout 900 python bench.py --cells 100 --steps 5 --warmup 2 2>$OUT/final_bench_cfg2.err | tail -1 ) > $OUT/final_bench_cfg2_100cube.json
( timeout 1500 python bench.py --gpus 8 --same-device --backend gloo --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step 2>$OUT/final_bench_8ranks.err | tail -1 ) > $OUT/final_bench_cfg5_8ranks_sharing_one_gpu_gloo.json
( timeout 1500 python bench.py --gpus 8 --same-device --backend gloo --workload beam --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step --no-parity-step 2>$OUT/final_bench_beam8.err | tail -1 ) > $OUT/final_bench_beam_8ranks_sharing_one_gpu_gloo.json
( timeout 1500 python bench.py --gpus 2 --same-device --backend gloo --steps 2 --warmup 1 --no-jacobi-step 2>$OUT/final_bench_2ranks.err | tail -1 ) > $OUT/final_bench_2ranks_same_device_transports_ab.json
( timeout 900 python bench.py --jitter 0.2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>$OUT/final_bench_jitter.err | tail -1 ) > $OUT/final_bench_cfg3_jitter.json
( timeout 900 python bench.py --workload beam --jitter 0.2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-jacobi-step 2>$OUT/final_bench_beam_jitter.err | tail -1 ) > $OUT/final_bench_cfg4_jitter.json
( timeout 900 python bench.py --numbering rcb8 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-jacobi-step 2>$OUT/final_bench_rcb8.err | tail -1 ) > $OUT/final_bench_numbering_rcb8.json
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/final_prof_stats.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats 45 > $OUT/final_rocprofv3_kernel_stats.txt 2>&1
python tools/trace_gaps.py /tmp/prof_stats k_pc_update > $OUT/final_kernel_timeline_gamg_loop.txt 2>&1
python tools/trace_phase.py /tmp/prof_stats > $OUT/final_kernel_phases_of_a_step.txt 2>&1
rm -rf /tmp/prof_stats_j
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_j -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --pc jacobi --no-pmc > $OUT/final_prof_stats_j.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_j > $OUT/final_rocprofv3_kernel_stats_jacobi_loop.txt 2>&1
rm -rf /tmp/prof_stats_b
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_b -- python3 bench.py --workload beam --steps 3 --warmup 1 --no-jacobi-step --no-pmc > $OUT/final_prof_stats_b.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_b 40 > $OUT/final_rocprofv3_kernel_stats_beam.txt 2>&1
RE="k_spmv|k_cg_|k_pc_|k_amg_spmv|k_amg_cheb|k_amg_restrict|k_amg_prolong|k_amg_galerkin|k_lat_galerkin|k_amg_diag|k_amg_tail|k_gather"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "$RE" -f csv -d /tmp/prof_$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-pmc > $OUT/final_pmc_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_$C $C > $OUT/final_rocprofv3_pmc_$C.txt 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_b$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_spmvg" -f csv -d /tmp/prof_b$C -- python3 bench.py --workload beam --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > $OUT/final_pmc_beam_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_b$C $C > $OUT/final_rocprofv3_pmc_beam_$C.txt 2>&1
done
# SQ counters of the assembly kernel (what bounds it: VALU issue) and of the dictionary SpMV
: > $OUT/final_gather_and_spmv_sq_counters.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_k
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "k_gather_poisson_tet4|k_spmvr_vd<false" -f csv -d /tmp/pmc_k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step --no-pmc > /tmp/pmc_k.log 2>&1
  for c in $set; do python3 tools/summarize_prof.py pmc /tmp/pmc_k $c 2>/dev/null | tail -n +2 | head -2 | awk -v c=$c '{print c, $0}' | cut -c1-200; done
done >> $OUT/final_gather_and_spmv_sq_counters.txt 2>&1
timeout 600 python tools/probe_coupled.py 200 30 2>$OUT/final_probe_coupled.err | grep "^{" | tail -1 > $OUT/final_coupled_cycle_rccl_self_peer.json
timeout 1500 python tools/probe_partition.py 160 3 2>$OUT/final_partition.err | grep "^{" | tail -1 > $OUT/final_partition_stairs_160cube_3ranks.json
for f in n1 n1_fp64_values beam cfg5_single_gpu cfg2_100cube cfg5_8ranks_sharing_one_gpu_gloo beam_8ranks_sharing_one_gpu_gloo 2ranks_same_device_transports_ab cfg3_jitter cfg4_jitter numbering_rcb8; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/final_bench_$f.json"))
    print("$f", {k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, "jacobi", (d.get("jacobi_step") or {}).get("ms_per_step"), (d.get("jacobi_step") or {}).get("iterations"), "roof", round(d["roofline"]["frac"],3), round(d["roofline"].get("algorithmic_frac", 0),3), round(d["roofline"]["avg_launch_ms"],4), d["preconditioner"].get("rows_per_level"), d["preconditioner"].get("symbolic_setup_ms_once_per_pattern"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print("$f", "ERR", e)
PY
done
head -14 $OUT/final_rocprofv3_kernel_stats.txt; head -3 $OUT/final_kernel_timeline_gamg_loop.txt; cat $OUT/final_gather_and_spmv_sq_counters.txt | head -30
python3 -c "
import json
d=json.load(open('$OUT/final_coupled_cycle_rccl_self_peer.json'))
for k,r in d.items():
    if isinstance(r, dict) and 'ms_per_iteration' in r: print(k, round(r['ms_per_iteration'],3), r['iterations'], round(r['symbolic_setup_ms'],2), r['distributed_levels'])
d=json.load(open('$OUT/final_partition_stairs_160cube_3ranks.json'))
for k,v in d.items():
    if isinstance(v, dict): print(k, v['iterations'], v['aggregation'], v['symbolic_ms_per_rank'], v['rows_per_level_owned_by_rank'][0])
"
