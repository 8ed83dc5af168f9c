#!/bin/bash
# 16-bit gaps with escapes against int32 columns, same box: config 3's mesh under the reference's 8-part renumbering and under a
# random node permutation (internal Morton order), Jacobi loop (the SpMV's share is largest there)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out/escape_ab.txt
: > $OUT
for NB in rcb8 shuffle; do for E in 0 1; do
  X=""; [ $E = 0 ] && X="PFEM_DEBUG_NO_GAP_ESCAPES=1"
  line=$( env $X A=1 timeout 900 python bench.py --numbering $NB --pc jacobi --steps 3 --warmup 1 --no-cpu-baseline --no-parity-step 2>/dev/null | tail -1 )
  python3 - "$NB escapes=$E" "$line" >> $OUT <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print(f"{sys.argv[1]:22s} step {d['ms_per_step']:7.2f} ms  its {d['iterations']}  per-it {d['ms_per_iteration']:.4f}  spmv {r['avg_launch_ms']*1e3:.1f} us  frac {r['frac']:.3f}  {r['kernel'][:32]}")
PY
done; done
cat $OUT
