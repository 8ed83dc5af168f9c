#!/bin/bash
# GPU visit: dictionary form of the relative-row-group gap stream -- parity tests, then A/B against the 32-bit gap form
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "relative_row_groups or spmv" 2>&1 | tail -8 ) > $OUT/pytest_dict.log 2>&1
tail -4 $OUT/pytest_dict.log
for g32 in 1 0 1 0; do
  if [ $g32 = 1 ]; then export PFEM_DEBUG_REL_GAP32=1; else unset PFEM_DEBUG_REL_GAP32; fi
  ( timeout 300 python tools/probe_slab_spmv.py 2>/dev/null | grep '^{' | tail -1 ) > $OUT/slab_spmv_gap32_$g32.json
  python - "$OUT/slab_spmv_gap32_$g32.json" $g32 <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("forced_gap32", sys.argv[2], [(f, [(r["column_bits"], r["us"], r["form_bytes"]) for r in rs]) for f, rs in d["forms"].items()])
PY
done
unset PFEM_DEBUG_REL_GAP32
( timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline --no-parity-step 2>/dev/null | grep "^{" | tail -1 ) > $OUT/bench_cfg5_single_dict.json
python - "$OUT/bench_cfg5_single_dict.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print({k:d.get(k) for k in ("value","ms_per_step","iterations","ms_per_iteration")}, d["roofline"]["kernel"][:40], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
