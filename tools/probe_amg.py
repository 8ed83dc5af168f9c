#!/usr/bin/env python3
"""-pc_type gamg against the oracle's restatement and against point Jacobi: iteration counts, times, hierarchy.
    python tools/probe_amg.py [cells ...]          (Poisson cubes; `beam:S` = the beam at scale S/10)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H


def run(spec, check_oracle):
    beam = spec.startswith("beam:")
    if beam:
        sc = int(spec.split(":")[1])
        nE = (5 * sc, 30 * sc, 5 * sc); ext = (-0.5, 0.5, 0.0, 6.0, -0.5, 0.5); kind = pf.ELAST_TET; ndof = 3; bc = 1; ed = H.ELAST_ELEMDATA
    else:
        n = int(spec); nE = (n, n, n); ext = (-1, 1, -1, 1, -1, 1); kind = pf.POISSON_TET; ndof = 1; bc = 0; ed = H.POISSON_ELEMDATA
    box = (ext[0], ext[1], nE[0], ext[2], ext[3], nE[1], ext[4], ext[5], nE[2])
    sz = H.box_slab_sizes(*nE, bc, ndof)
    out = {"case": spec, "free_dofs": sz["size_global"]}
    for pc in ("jacobi", "gamg"):
        s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"])
        s.setTolerances(rtol=1e-5, maxits=100000)
        s.setPreconditioner(pc)
        s.generateBoxMesh(kind, *box, bc_mode=bc)
        s.buildPattern()
        best = None
        for rep in range(3):
            s.assemble(ed, H.TIMEDATA)
            t0 = time.perf_counter()
            its, reason, rn = s.factoriseAndSolve()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        x = s.getSolution()
        out[pc] = {"its": its, "reason": reason, "solve_ms": best * 1e3, "pc_in_effect": s.preconditioner()}
        if pc == "gamg":
            info = s.amgInfo()
            out[pc]["hierarchy"] = info
            if check_oracle:
                from oracle import pfem_oracle as O
                rowptr, cols, vals = s.getCSR()
                rhs = s.getRHS()
                aggs = [s.amgAggregates(l, info["rows"][l]) for l in range(info["levels"] - 1)]
                sizes = [np.bincount(np.bincount(a)).tolist() for a in aggs]
                xo, ito, ro, rno, hist = O.pcg_amg(rowptr, cols, vals, rhs, aggs, cheb_degree=info["cheb_degree"], fine_degree=info["fine_degree"], eig_ratio=info["eig_ratio"],
                                                   coarse_scale=info["coarse_scale"])
                out[pc]["oracle"] = {"its": ito, "reason": ro, "max_abs_diff": float(np.abs(x - xo).max()), "aggregate_size_histograms": sizes,
                                     "history_rel_diff": float(np.abs(s.getHistory()[:len(hist)] - hist[:len(s.getHistory())]).max() / hist[0])}
        else:
            xj = x
        s.free()
    out["max_abs_diff_gamg_vs_jacobi_solution"] = None
    return out


if __name__ == "__main__":
    specs = sys.argv[1:] or ["20", "40"]
    for sp_ in specs:
        big = (not sp_.startswith("beam:") and int(sp_) > 64) or (sp_.startswith("beam:") and int(sp_.split(":")[1]) > 4)
        print(json.dumps(run(sp_, check_oracle=not big)), flush=True)
