import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pfemfort_amd as pf
from pfemfort_amd import host as H
n=200
for rep in range(3):
    t0=time.perf_counter(); sz=H.box_slab_sizes(n,n,n); s=pf.PetscSolver().initialise(sz["size_local"], sz["size_global"]); t1=time.perf_counter()
    s.generateBoxMesh(pf.POISSON_TET,-1.0,1.0,n,-1.0,1.0,n,-1.0,1.0,n); t2=time.perf_counter()
    s.buildPattern(); t3=time.perf_counter()
    print(f"rep {rep}: create {t1-t0:.3f} generate {t2-t1:.3f} (upload_ms {s.timings()['upload_ms']:.1f}) pattern {t3-t2:.3f} (pattern_ms {s.timings()['pattern_ms']:.1f})")
    s.free()
