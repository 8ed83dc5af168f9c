"""Iteration counts of CG with the reference's own preconditioner -- PCBJACOBI with its default ILU(0) sub-solver,
one block per rank (solverpetsc.F:187, 206) -- against point Jacobi, from the oracle's restatement on the CPU
(oracle/pfem_oracle.c: orc_pcg_bjacobi_ilu0).  Input of the "is ILU(0) worth building on the GPU" decision
(profiles/r02/ilu0_*).   python tools/probe_ilu0_iterations.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfem_oracle as O   # noqa: E402


def case(name, kind, mesh, blocks_list, rtol=1e-5, maxits=100000):
    prob = O.setup_problem(kind, mesh)
    N = prob.dm.size_global
    out = {"case": name, "free_dofs": int(N), "nnz": int(len(prob.cols)), "rtol": rtol}
    O.set_threads(os.cpu_count() or 1)
    t = time.time()
    _, its, reason, _, _ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=rtol, maxits=maxits)
    out["jacobi"] = {"its": its, "reason": reason, "cpu_s": time.time() - t}
    for nb in blocks_list:
        bs = [N * b // nb for b in range(nb + 1)]
        t = time.time()
        _, its, reason, _ = O.pcg_bjacobi_ilu0(prob.rowptr, prob.cols, prob.vals, prob.rhs, block_start=bs, rtol=rtol, maxits=maxits)
        out[f"bjacobi_ilu0_{nb}_blocks"] = {"its": its, "reason": reason, "cpu_s": time.time() - t}
        print(name, nb, out[f"bjacobi_ilu0_{nb}_blocks"], out["jacobi"], flush=True)
    return out


def main():
    res = []
    for n in (50, 100, 200):
        res.append(case(f"poisson_{n}^3", O.POISSON_TET, O.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n), [1, 8]))
        json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/ilu0_its.json", "w"), indent=1)
    res.append(case("beam_50x300x50", O.ELAST_TET, O.gen_box_tets(-0.5, 0.5, 50, 0.0, 6.0, 300, -0.5, 0.5, 50, bc_mode=1, ndof=3), [1, 8]))
    json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/ilu0_its.json", "w"), indent=1)


if __name__ == "__main__":
    main()
