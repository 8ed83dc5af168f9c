"""SpMV variant lab on the real 200^3 matrix (development aid; uses libpfem_amd_lab.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libpfem_amd.so", "libpfem_amd_lab.so")
import numpy as np
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern(); s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
info = s.matrixInfo(); byts = 12 * info["nnz"] + 20 * info["n_local"]
lab = L.lib().pfem_lab_spmv
lab.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
def run(var, grid, block, reps=30):
    ms = C.c_double(0); chk = C.c_double(0)
    rc = lab(s._h, var, grid, block, reps, C.byref(ms), C.byref(chk))
    return rc, ms.value, chk.value
names = {0: "int32 product", 13: "16-bit product", 12: "16-bit 2rows/lane", 11: "int32 2rows nt"}
ns = info["n_local"] // 64 + 1
for rep in range(3):
    for var in (0, 13, 12, 11):
        grid = (ns // 8 + 1) if var == 11 else 1
        rc, ms, chk = run(var, grid, 256, reps=50)
        print(f"var {var:2d} {names[var]:20s} rc {rc} {ms*1e3:8.1f} us  {byts/ms/1e6:7.0f} GB/s  chk {chk:.6e}", flush=True)
