#!/bin/bash
# the pattern from the incidence lists (default) against the sorted element-matrix keys (PFEM_DEBUG_PATTERN_SORT=1):
# build times at 200^3 / 400^3 / the beam, then the whole GPU suite
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
cat > /tmp/pat.py <<'PY'
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pfemfort_amd as pf
from pfemfort_amd import host as H
n=int(sys.argv[1])
for rep in range(int(sys.argv[2])):
    t0=time.perf_counter(); sz=H.box_slab_sizes(n,n,n); s=pf.PetscSolver().initialise(sz["size_local"], sz["size_global"]); t1=time.perf_counter()
    s.generateBoxMesh(pf.POISSON_TET,-1.0,1.0,n,-1.0,1.0,n,-1.0,1.0,n); t2=time.perf_counter()
    s.buildPattern(); t3=time.perf_counter()
    print(f"n={n} rep {rep}: create {t1-t0:.3f} generate {t2-t1:.3f} pattern {t3-t2:.3f} s (pattern_ms on the stream {s.timings()['pattern_ms']:.1f})", flush=True)
    s.free()
PY
for S in 0 1; do
  if [ $S = 1 ]; then export PFEM_DEBUG_PATTERN_SORT=1; else unset PFEM_DEBUG_PATTERN_SORT; fi
  echo "== sorted keys: $S"
  timeout 600 python /tmp/pat.py 200 3 2>&1 | grep "n="
  timeout 600 python /tmp/pat.py 400 2 2>&1 | grep "n="
done
unset PFEM_DEBUG_PATTERN_SORT
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pattern_lists_parity.log 2>&1
tail -5 $OUT/pattern_lists_parity.log
if grep -q "failed\|error" $OUT/pattern_lists_parity.log; then exit 1; fi
bash tools/gpu_suite.sh lists
