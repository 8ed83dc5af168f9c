"""A/B of the hipGraph replay of the CG iteration (PFEM_CG_GRAPH is read once per process: two subprocesses)."""
import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
n = int(sys.argv[1])
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern()
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
for prof in (0, 8):
    s.profileSpmv(prof)
    for rep in range(3):
        its, reason, rn = s.factoriseAndSolve(); tm = s.timings()
    u = s.getSolution()
    print(f"graph={os.environ['PFEM_CG_GRAPH']} profile={prof}: solve {tm['solve_ms']:.2f} ms its {its} reason {reason} per-iter {tm['solve_ms']/its*1e3:.1f} us sum(u)={u.sum():.15e} hist_last={s.getHistory()[-1]:.6e}")
'''
for n in sys.argv[1:] or ["200"]:
    for g in ("0", "1"):
        subprocess.run([sys.executable, "-c", code, n], env=dict(os.environ, PFEM_CG_GRAPH=g))
