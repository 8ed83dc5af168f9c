#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_lean3.log 2>&1
tail -5 $OUT/pytest_lean3.log
for w in poisson beam; do
for plain in 1 0 1 0; do
  if [ $plain = 1 ]; then export PFEM_DEBUG_GATHER_PLAIN_ORDER=1; else unset PFEM_DEBUG_GATHER_PLAIN_ORDER; fi
  ( timeout 600 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline --no-parity-step 2>$OUT/bench_lean.err | grep '^{' | tail -1 ) > $OUT/bench_lean3_${w}_plain$plain.json
  python - "$OUT/bench_lean3_${w}_plain$plain.json" $w $plain <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], "plain_order", sys.argv[3], {k:d.get(k) for k in ("ms_per_step","assembly_ms_per_step","solve_ms_per_step")})
PY
done
done
unset PFEM_DEBUG_GATHER_PLAIN_ORDER
rm -rf /tmp/prof_fetch
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_gather" -f csv -d /tmp/prof_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-step > $OUT/prof_fetch_lean3.log 2>&1
python tools/summarize_prof.py pmc /tmp/prof_fetch FETCH_SIZE
rm -rf /tmp/prof_stats
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/prof_stats_lean3.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats | grep -E "kernel|k_gather"
