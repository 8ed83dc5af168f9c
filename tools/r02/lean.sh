#!/bin/bash
# GPU visit: lean element geometry in the gather kernels -- parity (assembled K, F bit-exact vs the oracle) and timing
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_lean.log 2>&1
tail -5 $OUT/pytest_lean.log
for w in poisson beam; do
  ( timeout 600 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline --no-parity-step 2>$OUT/bench_lean.err | grep '^{' | tail -1 ) > $OUT/bench_lean_$w.json
  python - "$OUT/bench_lean_$w.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("value","ms_per_step","iterations","assembly_ms_per_step","solve_ms_per_step")})
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_lean -o lean -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step > $OUT/prof_lean.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/prof_lean/**/*kernel_stats.csv", recursive=True)
for p in f:
    rows=list(csv.DictReader(open(p)))
    for r in rows[:12]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
