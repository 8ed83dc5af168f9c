import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern()
for i in range(2):
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
print(s.timings()["assemble_ms"])
