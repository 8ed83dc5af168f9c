"""Do idle OpenMP workers slow the launch-bound GPU loop that follows a host bookkeeping call?  A small Jacobi solve is timed
right after pfem_gen_box_tets (OpenMP loops on the host) and again after a pause.
usage: [OMP_WAIT_POLICY=active KMP_BLOCKTIME=200] lab_omp_spin.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H

n = 100
sz = H.box_slab_sizes(n, n, n)
s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"])
s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, 1.0, n)
s.buildPattern()
s.setPreconditioner("jacobi")
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
for _ in range(3):
    s.factoriseAndSolve()


def solve_ms():
    t0 = time.perf_counter(); s.factoriseAndSolve(); return (time.perf_counter() - t0) * 1e3


time.sleep(1.0)
quiet = [solve_ms() for _ in range(5)]
H.gen_box_tets(-1, 1, 60, -1, 1, 60, -1, 1, 60)          # OpenMP loops
after = [solve_ms() for _ in range(5)]
print("OMP_WAIT_POLICY", os.environ.get("OMP_WAIT_POLICY"), "KMP_BLOCKTIME", os.environ.get("KMP_BLOCKTIME"))
print("solve ms, quiet:                ", " ".join(f"{v:6.2f}" for v in quiet))
print("solve ms, after a host OMP call:", " ".join(f"{v:6.2f}" for v in after))
