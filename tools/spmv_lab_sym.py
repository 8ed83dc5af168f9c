"""Lab: would a symmetric-half storage of the relative-row-group SpMV pay?  (timing/traffic emulation, variant 20 of
pfem_lab.inc; uses libpfem_amd_lab.so)"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd._lib as L
L.LIB_PATH = L.LIB_PATH.replace("libpfem_amd.so", "libpfem_amd_lab.so")
import pfemfort_amd as pf
from pfemfort_amd import host as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sz = H.box_slab_sizes(n, n, n)
s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"])
s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, 1.0, n)
s.buildPattern(); s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
s.setTolerances(rtol=1e-5, maxits=3); s.factoriseAndSolve()          # builds the relative-row-group copy
info = s.matrixInfo(); byts = 12 * info["nnz"] + 20 * info["n_local"]
lab = L.lib().pfem_lab_spmv
lab.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
out = {}
for rep in range(3):
    for var, name in ((13, "product_k_spmvr"), (20, "symmetric_half_emulation")):
        ms = C.c_double(0); chk = C.c_double(0)
        rc = lab(s._h, var, 1, 256, 50, C.byref(ms), C.byref(chk))
        out.setdefault(name, []).append(round(ms.value * 1e3, 1))
print(json.dumps({"cells": n, "us_per_launch": out, "algorithmic_bytes": byts}))
