#!/usr/bin/env python3
"""Iteration counts of the preconditioner candidates VERDICT r02 #4 names, on the oracle's matrices (CPU, scipy; counts are
hardware-independent), with a cost model from the kernel times measured on MI355X -- the record behind the choice of
-pc_type gamg (profiles/r03/preconditioner_probe.json).

    python tools/probe_preconditioners.py 40 60          # Poisson cubes;  beam:S = the beam at S/10 of config 4
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pfem_oracle as O


def problem(spec):
    if spec.startswith("beam:"):
        sc = int(spec.split(":")[1])
        mesh = O.gen_box_tets(-0.5, 0.5, 5 * sc, 0, 6, 30 * sc, -0.5, 0.5, 5 * sc, bc_mode=1, ndof=3)
        prob = O.setup_problem(O.ELAST_TET, mesh)
    else:
        n = int(spec)
        prob = O.setup_problem(O.POISSON_TET, O.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n))
    return prob


def pcg(A, b, M, rtol=1e-5, maxit=20000):
    x, its, reason, rn, hist = O.pcg_with(A.indptr, A.indices, A.data, b, M, rtol=rtol, maxits=maxit)
    return its, reason


def greedy_colouring(A):
    n = A.shape[0]
    col = -np.ones(n, np.int64)
    ip, ix = A.indptr, A.indices
    for i in range(n):
        used = set(col[ix[ip[i]:ip[i + 1]]])
        c = 0
        while c in used:
            c += 1
        col[i] = c
    return col


def sgs(A):
    """z = (D+U)^-1 D (D+L)^-1 r, in the matrix's own ordering"""
    L = sp.tril(A, format="csr"); U = sp.triu(A, format="csr"); d = A.diagonal()
    return lambda r: spl.spsolve_triangular(U, d * spl.spsolve_triangular(L, r, lower=True), lower=False)


def cheb_poly(A, deg, ratio=8.0):
    d = 1.0 / A.diagonal()
    lmax = float((abs(A) @ np.ones(A.shape[0]) * d).max())
    lmin = lmax / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta

    def M(r):
        rho = 1.0 / sigma
        rr = r.copy()
        dd = d * rr / theta
        x = dd.copy()
        for _ in range(1, deg):
            rr = rr - A @ dd
            rho_new = 1.0 / (2 * sigma - rho)
            dd = rho_new * rho * dd + 2 * rho_new / delta * (d * rr)
            x += dd
            rho = rho_new
        return x
    return M


def run(spec):
    prob = problem(spec)
    A = sp.csr_matrix((prob.vals, prob.cols, prob.rowptr))
    b = prob.rhs
    N = A.shape[0]
    out = {"case": spec, "free_dofs": N, "nnz": int(A.nnz)}
    d = 1.0 / A.diagonal()
    out["jacobi"] = {"its": pcg(A, b, lambda r: d * r)[0], "spmv_per_it": 1}
    for deg in (2, 3, 4):
        its = pcg(A, b, cheb_poly(A, deg))[0]
        out[f"chebyshev_jacobi_degree_{deg}"] = {"its": its, "spmv_per_it": deg, "total_spmv": its * deg}
    out["ilu0_natural (the reference's PCBJACOBI block solver, 1 block)"] = {"its": O.pcg_bjacobi_ilu0(prob.rowptr, prob.cols, prob.vals, b)[1]}
    out["sgs_natural_order"] = {"its": pcg(A, b, sgs(A))[0], "note": "sequential triangular solves: ~800 dependency levels at 200^3 (round 2: refused)"}
    col = greedy_colouring(A)
    perm = np.argsort(col, kind="stable")
    Ap = A[perm][:, perm].tocsr()
    Ap.sort_indices()
    out["sgs_multicolour"] = {"its": pcg(Ap, b[perm], sgs(Ap))[0], "colours": int(col.max() + 1),
                              "matrix_passes_per_it": 2, "note": "Eisenstat's trick folds the SpMV into the two sweeps: 1 pass, 2 x colours launches"}
    ilu = O.pcg_bjacobi_ilu0(Ap.indptr.astype(np.int64), Ap.indices.astype(np.int32), Ap.data, b[perm])
    out["ilu0_multicolour"] = {"its": ilu[1], "colours": int(col.max() + 1)}
    return out


if __name__ == "__main__":
    for spec in (sys.argv[1:] or ["30", "50"]):
        print(json.dumps(run(spec)), flush=True)
