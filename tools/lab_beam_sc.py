"""scatter against gather assembly on the full-size beam (diagnostic of a failing test)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nx, ny = max(2, round(50 * scale)), max(4, round(300 * scale))
mesh = H.gen_box_tets(-0.5, 0.5, nx, 0.0, 6.0, ny, -0.5, 0.5, nx, bc_mode=1, ndof=3)
dm, conn, xyz, edof = D._setup(pf.ELAST_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.ELAST_TET, conn, xyz, edof, dm.solnApplied)
s.buildPattern()
print("info", s.matrixInfo(), s.assemblyInfo() if hasattr(s, "assemblyInfo") else None)
res = {}
for mode in ("scatter", "gather", "scatter", "gather"):
    s.setAssemblyMode(mode); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    rowptr, cols, v = s.getCSR()
    f = s.getRHS()
    if mode in res:
        print(mode, "again: equal to its first", np.array_equal(res[mode][0], v), np.array_equal(res[mode][1], f))
    res[mode] = (v.copy(), f.copy())
v_sc, v_g = res["scatter"][0], res["gather"][0]
bad = np.nonzero(np.abs(v_sc - v_g) > 1e-12 * np.abs(v_g).max())[0]
print("entries that differ:", len(bad), "of", len(v_g))
if len(bad):
    rows = np.searchsorted(rowptr, bad, side="right") - 1
    print("rows:", rows[:20], "...", rows[-5:], "distinct rows", len(np.unique(rows)))
    print("first:", [(int(b), float(v_sc[b]), float(v_g[b])) for b in bad[:8]])
    print("rhs differ:", int((np.abs(res["scatter"][1] - res["gather"][1]) > 1e-12).sum()))
