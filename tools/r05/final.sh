#!/bin/bash
# round 5, evidence at HEAD: driver-form bench lines (N=1 default, beam, config 5 alone, 100^3, 8 ranks sharing the GPU for
# config 5), rocprofv3 kernel stats (default bench, Jacobi loop, beam) and the two PMC passes of the default bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/final_bench_n1.err | tail -1 ) > $OUT/final_bench_n1.json
( timeout 900 python bench.py --workload beam --steps 5 --warmup 2 2>$OUT/final_bench_beam.err | tail -1 ) > $OUT/final_bench_beam.json
( PFEM_SPMV_VALDICT=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/final_bench_n1_fp64_values.err | tail -1 ) > $OUT/final_bench_n1_fp64_values.json
( PFEM_SPMV_VALDICT=0 timeout 900 python bench.py --workload beam --steps 5 --warmup 2 --no-cpu-baseline 2>$OUT/final_bench_beam_fp64_values.err | tail -1 ) > $OUT/final_bench_beam_fp64_values.json
( timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline 2>$OUT/final_bench_cfg5.err | tail -1 ) > $OUT/final_bench_cfg5_single_gpu.json
( timeout 900 python bench.py --cells 100 --steps 5 --warmup 2 2>$OUT/final_bench_cfg2.err | tail -1 ) > $OUT/final_bench_cfg2_100cube.json
( timeout 1500 python bench.py --gpus 8 --same-device --backend gloo --steps 1 --warmup 1 --no-transport-ab --no-jacobi-step 2>$OUT/final_bench_8ranks.err | tail -1 ) > $OUT/final_bench_cfg5_8ranks_sharing_one_gpu_gloo.json
rm -rf /tmp/prof_stats
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/final_prof_stats.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats 40 > $OUT/final_rocprofv3_kernel_stats.txt 2>&1
python tools/trace_gaps.py /tmp/prof_stats k_pc_update > $OUT/final_kernel_timeline_gamg_loop.txt 2>&1
rm -rf /tmp/prof_stats_j
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_j -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step --pc jacobi > $OUT/final_prof_stats_j.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_j > $OUT/final_rocprofv3_kernel_stats_jacobi_loop.txt 2>&1
rm -rf /tmp/prof_stats_b
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_b -- python3 bench.py --workload beam --steps 3 --warmup 1 --no-jacobi-step > $OUT/final_prof_stats_b.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_b 40 > $OUT/final_rocprofv3_kernel_stats_beam.txt 2>&1
RE="k_spmv|k_cg_|k_pc_|k_amg_spmv|k_amg_cheb|k_amg_restrict|k_amg_prolong|k_amg_galerkin|k_lat_galerkin|k_amg_diag|k_amg_tail|k_gather"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "$RE" -f csv -d /tmp/prof_$C -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/final_pmc_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_$C $C > $OUT/final_rocprofv3_pmc_$C.txt 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_b$C
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "k_spmvg" -f csv -d /tmp/prof_b$C -- python3 bench.py --workload beam --steps 1 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/final_pmc_beam_$C.log 2>&1
  python tools/summarize_prof.py pmc /tmp/prof_b$C $C > $OUT/final_rocprofv3_pmc_beam_$C.txt 2>&1
done
for f in n1 n1_fp64_values beam beam_fp64_values cfg5_single_gpu cfg2_100cube cfg5_8ranks_sharing_one_gpu_gloo; do python3 - <<PY
import json
try:
    d=json.load(open("$OUT/final_bench_$f.json"))
    print("$f", {k:d.get(k) for k in ("value","cold_value","ms_per_step","iterations","ms_per_iteration","assembly_ms_per_step","first_step_ms_including_once_per_pattern_setup")}, "jacobi", (d.get("jacobi_step") or {}).get("ms_per_step"), (d.get("jacobi_step") or {}).get("iterations"), "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_ms"],4), d["preconditioner"].get("rows_per_level"), d["preconditioner"].get("symbolic_setup_ms_once_per_pattern"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("host_stream_triad_gbps"))
except Exception as e: print("$f", "ERR", e)
PY
done
head -16 $OUT/final_rocprofv3_kernel_stats.txt; grep -E "k_spmvr|k_spmvg|k_gather|k_amg_spmv_ep<0>|k_lat_galerkin" $OUT/final_rocprofv3_pmc_FETCH_SIZE.txt $OUT/final_rocprofv3_pmc_WRITE_SIZE.txt $OUT/final_rocprofv3_pmc_beam_FETCH_SIZE.txt $OUT/final_rocprofv3_pmc_beam_WRITE_SIZE.txt
