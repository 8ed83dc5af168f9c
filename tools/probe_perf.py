"""Quick single-GPU probe: sizes, phase times, SpMV GB/s (development aid, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
print(pf.device_info(0))
t = time.time(); mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n); print(f"gen {time.time()-t:.2f}s nNode={mesh.nNode} nElem={mesh.nElem}")
t = time.time(); dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh); print(f"bookkeeping {time.time()-t:.2f}s N={dm.size_global}")
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.setTolerances(rtol=rtol)
t = time.time(); s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); print(f"upload {time.time()-t:.2f}s")
t = time.time(); s.buildPattern(); print(f"pattern {time.time()-t:.2f}s", s.matrixInfo(), "column bits", s.spmvColumnBits())
for rep in range(3):
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    s.profileSpmv(True)
    its, reason, rn = s.factoriseAndSolve()
    tm = s.timings()
    info = s.matrixInfo()
    byts = 12 * info["nnz"] + 20 * info["n_local"]
    spmv_ms = tm["spmv_ms_total"] / max(tm["spmv_launches"], 1)
    print(f"rep{rep}: assemble {tm['assemble_ms']:.2f} ms  solve {tm['solve_ms']:.1f} ms its={its} reason={reason} rn={rn:.3e} "
          f"spmv {spmv_ms:.4f} ms -> {byts/spmv_ms/1e6:.0f} GB/s ({byts/spmv_ms/1e6/8000*100:.1f}% of 8 TB/s)  "
          f"DOF/s={dm.size_global/((tm['assemble_ms']+tm['solve_ms'])/1e3):.3e}  per-iter {tm['solve_ms']/max(its,1):.3f} ms")
ms = s.benchSpmv(50)
print(f"standalone spmv {ms:.4f} ms -> {byts/ms/1e6:.0f} GB/s")
u = s.getSolution()
full = dm.solnApplied.copy(); full[H.assy_for_soln(dm.NodeDofArrayNew)] = u
print("max err vs x^2+y^2+z^2:", np.abs(full - (mesh.xyz**2).sum(0)).max())
