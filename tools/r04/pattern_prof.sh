#!/bin/bash
# kernel trace of mesh generation + pattern build at 200^3 (three builds in one process) and the new parity test
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
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
rm -rf /tmp/prof_pat
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_pat -- python3 /tmp/pat.py 200 3 > $OUT/pattern_prof.log 2>&1
grep "n=" $OUT/pattern_prof.log
python tools/summarize_prof.py stats /tmp/prof_pat 40 > $OUT/pattern_kernel_stats.txt 2>&1
cat $OUT/pattern_kernel_stats.txt | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pattern" 2>&1 | tail -3
