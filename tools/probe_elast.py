"""cfg 4 probe: 50x300x50x6 beam (or scaled), elasticity, one GPU (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
nx, ny, nz = int(50 * sc), int(300 * sc), int(50 * sc)
t = time.time(); mesh = H.gen_box_tets(-0.5, 0.5, nx, 0.0, 6.0, ny, -0.5, 0.5, nz, bc_mode=1, ndof=3)
dm, conn, xyz, edof = D._setup(pf.ELAST_TET, mesh); print(f"setup {time.time()-t:.2f}s nodes {mesh.nNode} elems {mesh.nElem} N {dm.size_global}")
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global); s.setTolerances(rtol=rtol, maxits=100000)
s.uploadMesh(pf.ELAST_TET, conn, xyz, edof, dm.solnApplied)
t = time.time(); s.buildPattern(); print(f"pattern {time.time()-t:.2f}s", s.matrixInfo(), "column bits", s.spmvColumnBits(), "rows per lane", s.spmvRowGroup())
for mode in ("gather", "scatter"):
    s.setAssemblyMode(mode); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA); print(mode, "assemble ms", s.timings()["assemble_ms"])
s.setAssemblyMode("gather"); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
s.profileSpmv(True)
its, reason, rn = s.factoriseAndSolve(); tm = s.timings(); info = s.matrixInfo()
byts = 12 * info["nnz"] + 20 * info["n_local"]; sp = tm["spmv_ms_total"] / max(tm["spmv_launches"], 1)
print(f"solve {tm['solve_ms']:.1f} ms its {its} reason {reason} rn {rn:.3e} spmv {sp:.4f} ms {byts/sp/1e6:.0f} GB/s  DOF/s {dm.size_global/((tm['assemble_ms']+tm['solve_ms'])/1e3):.3e}")
u = s.getSolution(); full = dm.solnApplied.copy(); full[H.assy_for_soln(dm.NodeDofArrayNew)] = u
disp = np.linalg.norm(full.reshape(-1, 3), axis=1)
print("max displacement magnitude", disp.max(), " (docs image: 0.82; beam theory 0.81)")
s.setPreconditioner("pbjacobi"); print("pc in effect:", s.preconditioner())
its, reason, rn = s.factoriseAndSolve(); tm = s.timings()
print(f"pbjacobi: solve {tm['solve_ms']:.1f} ms its {its} reason {reason} rn {rn:.3e}")
u2 = s.getSolution(); print("max |u_pbjacobi - u_jacobi| =", np.abs(u2 - u).max())
