#!/bin/bash
# round 6: the rigid-body restriction with 16 lanes per coarse node (k_rbm_restrict_wide) against the one-thread form, on config 4
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "gamg or beam or elast or rbm" 2>&1 | tail -15 ) > $OUT/beam_parity.txt 2>&1
( timeout 1200 python -m pytest tests/test_distributed.py -x -q -m gpu -k "elast" 2>&1 | tail -15 ) > $OUT/beam_dist.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -k "beam or elast" 2>&1 | tail -8 ) > $OUT/beam_full.txt 2>&1
for W in 0 1; do
  ( PFEM_RBM_RESTRICT_WIDE=$W timeout 900 python bench.py --workload beam --steps 5 --warmup 2 --no-cpu-baseline --no-jacobi-step --no-pmc 2>$OUT/beam_w$W.err | tail -1 ) > $OUT/beam_w$W.json
done
rm -rf /tmp/prof_stats_b
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_stats_b -- python3 bench.py --workload beam --steps 3 --warmup 1 --no-jacobi-step --no-pmc --no-cpu-baseline > $OUT/beam_prof.log 2>&1
python tools/summarize_prof.py stats /tmp/prof_stats_b 40 > $OUT/beam_kernel_stats.txt 2>&1
tail -4 $OUT/beam_parity.txt $OUT/beam_dist.txt $OUT/beam_full.txt
for W in 0 1; do python3 - <<PY
import json
d=json.load(open("$OUT/beam_w$W.json")); print("wide=$W", d["ms_per_step"], d["iterations"], d.get("ms_per_iteration"), d["value"])
PY
done
head -30 $OUT/beam_kernel_stats.txt | cut -c1-160
