#!/bin/bash
# knob sweep on the config-4 beam with the rigid-body-mode coarse space: coarse-correction scale, smoothing interval, degree, aggregate size
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/beam_knobs.txt
: > $OUT
run() {
  local tag="$1"; shift
  local line
  line=$( env "$@" timeout 600 python bench.py --workload beam --steps 2 --warmup 1 --no-jacobi-step 2>/dev/null | tail -1 )
  python3 - "$tag" "$line" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    p = d["preconditioner"]
    print(f"{sys.argv[1]:42s} its {d['iterations']:4d}  step {d['ms_per_step']:7.2f} ms  per-it {d['ms_per_iteration']:.3f}  numeric {p['numeric_setup_ms_per_solve_inside_the_timer']:.2f}  cold {d['first_step_ms_including_once_per_pattern_setup']:.1f}  complexity {p['operator_complexity']:.2f} rows {p['rows_per_level']}")
except Exception as e:
    print(sys.argv[1], "ERR", e, sys.argv[2][:200])
PY
}
run "default" A=1
for sc in 1.0 1.25 1.8; do run "coarse_scale=$sc" PFEM_AMG_COARSE_SCALE=$sc; done
for r in 4 16 30; do run "eig_ratio=$r" PFEM_AMG_EIG_RATIO=$r; done
run "cheb_degree=1" PFEM_AMG_CHEB_DEGREE=1
run "cheb_degree=1 scale=1.0" PFEM_AMG_CHEB_DEGREE=1 PFEM_AMG_COARSE_SCALE=1.0
run "cheb_degree=3" PFEM_AMG_CHEB_DEGREE=3
run "fine_degree=2" PFEM_AMG_FINE_DEGREE=2
run "passes0=4" PFEM_AMG_PASSES0=4
run "passes0=4 ratio=16" PFEM_AMG_PASSES0=4 PFEM_AMG_EIG_RATIO=16
run "passes=4 (all levels)" PFEM_AMG_PASSES=4
run "tail_rows=256" PFEM_AMG_TAIL_ROWS=256
run "tail_rows=64" PFEM_AMG_TAIL_ROWS=64
run "no absorb (odd node alone)" PFEM_AMG_LATTICE_ABSORB=0
run "graph off" PFEM_CG_GRAPH=0
cat $OUT
