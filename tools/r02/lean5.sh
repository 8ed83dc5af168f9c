#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_lean5.log 2>&1
tail -3 $OUT/pytest_lean5.log
for rep in 1 2 3; do
  ( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-step 2>$OUT/bench_lean.err | grep '^{' | tail -1 ) > $OUT/bench_lean5_$rep.json
  python - "$OUT/bench_lean5_$rep.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("ms_per_step","assembly_ms_per_step","solve_ms_per_step")})
PY
done
