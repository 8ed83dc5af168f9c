"""A/B of an environment-switched solver option inside one process / one box: tools/probe_env.py VAR [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
var = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern()
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
ref = None
for rep in range(3):
    for val in ("0", "1"):
        os.environ[var] = val
        its, reason, rn = s.factoriseAndSolve(); tm = s.timings(); u = s.getSolution()
        if ref is None: ref = u
        print(f"{var}={val}: solve {tm['solve_ms']:.2f} ms its {its} reason {reason} per-iter {tm['solve_ms']/its*1e3:.1f} us max|du| {np.abs(u-ref).max():.2e}")
