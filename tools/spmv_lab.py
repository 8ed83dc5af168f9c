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
names = {0: "product", 3: "noXCD", 9: "noXCD+nt", 10: "2rows noXCD", 11: "2rows noXCD nt"}
ns = info["n_local"] // 64 + 1
for var in (0, 3, 9, 10, 11):
    for grid, block in ((2048, 256), (4096, 256), (8192, 256), (16384, 256), (ns // 4 + 1, 256), (ns // 2 + 1, 128), (ns + 1, 64), (ns // 8 + 1, 512), (ns // 16 + 1, 1024), (ns // 8 + 1, 256), (ns // 16 + 1, 256)):
        if var == 0 and (grid, block) != (2048, 256): continue
        if var >= 10: grid = (grid + 1) // 2
        rc, ms, chk = run(var, grid, block)
        print(f"var {var:2d} {names[var]:15s} grid {grid:6d} block {block:5d} rc {rc} {ms*1e3:8.1f} us  {byts/ms/1e6:7.0f} GB/s  chk {chk:.6e}", flush=True)
