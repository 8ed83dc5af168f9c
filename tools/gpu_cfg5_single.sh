#!/bin/bash
# GPU visit: config 5 (400^3 x 6 tets, 63.5 M dofs, 949 M nonzeros) ALONE on one MI355X -- the strong-scaling baseline
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "element_ranges" 2>&1 | tail -5 ) > $OUT/pytest_ranges.log 2>&1
tail -3 $OUT/pytest_ranges.log
( timeout 900 python bench.py --cells 400 --steps 2 --warmup 1 --no-cpu-baseline 2>$OUT/bench_cfg5_single.err | grep '^{' | tail -1 ) > $OUT/bench_cfg5_single_gpu.json
tail -5 $OUT/bench_cfg5_single.err
python - "$OUT/bench_cfg5_single_gpu.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print({k:d.get(k) for k in ("value","ms_per_step","iterations","assembly_ms_per_step","solve_ms_per_step","ms_per_iteration","setup_s_untimed","max_nodal_error")})
    print(d["roofline"]["kernel"][:60], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
    print(d.get("parity_tolerance_step"))
except Exception as e:
    print("no result", e)
PY
rocm-smi --showmeminfo vram 2>/dev/null | head -5
