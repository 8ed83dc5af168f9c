"""Solve time with and without per-SpMV event pairs (how much the instrument costs); development aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
n = 200
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern()
for prof in (False, True, False, True):
    s.profileSpmv(prof)
    for rep in range(3):
        s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
        its, reason, rn = s.factoriseAndSolve()
    tm = s.timings()
    print(f"profile_spmv={prof}: solve {tm['solve_ms']:.2f} ms its {its} -> {tm['solve_ms']/its*1e3:.1f} us/iter; assemble {tm['assemble_ms']:.2f}")
