"""scatter against gather assembly on the full-size beam, repeated (diagnostic of a test that failed once in a full-suite run:
tests/test_gpu_full_size.py::test_elasticity_beam_config4, scatter and gather differing beyond 1e-12)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nx, ny = max(2, round(50 * scale)), max(4, round(300 * scale))
mesh = H.gen_box_tets(-0.5, 0.5, nx, 0.0, 6.0, ny, -0.5, 0.5, nx, bc_mode=1, ndof=3)
dm, conn, xyz, edof = D._setup(pf.ELAST_TET, mesh)
bad_runs = 0
v_ref = None
for rep in range(loops):
    if rep % 5 == 0:          # a fresh solver (and pattern build) every few rounds
        s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
        s.uploadMesh(pf.ELAST_TET, conn, xyz, edof, dm.solnApplied)
        s.buildPattern()
    s.setAssemblyMode("scatter"); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    v_sc = s.getCSR()[2]
    s.setAssemblyMode("gather"); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    v_g = s.getCSR()[2]
    if v_ref is None:
        v_ref = v_g.copy()
    tol = 1e-12 * np.abs(v_g).max()
    d = np.abs(v_sc - v_g)
    n_bad = int((d > tol).sum())
    g_same = bool(np.array_equal(v_g, v_ref))
    if n_bad or not g_same:
        bad_runs += 1
        i = int(np.argmax(d))
        print(f"rep {rep}: {n_bad} entries differ (max {d.max():.3e} at {i}: scatter {v_sc[i]!r} gather {v_g[i]!r} reference gather {v_ref[i]!r}); gather equals the first gather: {g_same}", flush=True)
print(f"{loops} rounds, {bad_runs} bad")
