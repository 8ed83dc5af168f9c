#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gamg" 2>&1 | tail -15 ) > $OUT/r03k_gamg_tests.log 2>&1
: > $OUT/r03k_knobs.log
for fused in 1 0; do
  echo "== fused $fused" >> $OUT/r03k_knobs.log
  PFEM_AMG_FUSED=$fused timeout 300 python tools/probe_amg.py 100 200 beam:10 2>&1 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: print(ln[:300]); continue
    print(d['case'], 'jacobi', d['jacobi']['its'], round(d['jacobi']['solve_ms'],1), 'gamg', d['gamg']['its'], round(d['gamg']['solve_ms'],1), 'numeric_ms', round(d['gamg']['hierarchy']['numeric_ms'],2))
" >> $OUT/r03k_knobs.log
done
( timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q -k "plain" 2>&1 | tail -8 ) > $OUT/r03k_bench_tests.log 2>&1
rm -rf /tmp/prof_amg
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_amg -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity-step --no-jacobi-step > $OUT/r03k_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_amg > $OUT/r03k_rocprof_kernel_stats_gamg_bench.txt 2>&1
tail -8 $OUT/r03k_gamg_tests.log; cat $OUT/r03k_knobs.log; tail -6 $OUT/r03k_bench_tests.log; head -24 $OUT/r03k_rocprof_kernel_stats_gamg_bench.txt
