"""What the reference's own preconditioner (PCBJACOBI -> ILU(0), solverpetsc.F:187,206) would cost per application
on MI355X, measured with the vendor's production kernels (rocSPARSE csrilu0 + csrsv, tools/lab/ilu0_probe.hip),
next to this library's Jacobi-PCG iteration on the same assembled matrix.  Together with the oracle's iteration
counts (tools/probe_ilu0_iterations.py -> profiles/r02/ilu0_iteration_counts.json) this is the measured basis of the
"ILU(0) is not built" decision.

    python tools/probe_ilu0_gpu.py [poisson200|poisson100|beam] > out.json
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pfemfort_amd as pf   # noqa: E402
from pfemfort_amd import host as H   # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "poisson200"
    if what == "beam":
        kind, ndof, bc, ed = pf.ELAST_TET, 3, 1, H.ELAST_ELEMDATA
        box = (-0.5, 0.5, 50, 0.0, 6.0, 300, -0.5, 0.5, 50)
    else:
        n = int(what.replace("poisson", ""))
        kind, ndof, bc, ed = pf.POISSON_TET, 1, 0, H.POISSON_ELEMDATA
        box = (-1.0, 1.0, n, -1.0, 1.0, n, -1.0, 1.0, n)
    sz = H.box_slab_sizes(box[2], box[5], box[8], bc, ndof)
    s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"])
    s.setTolerances(rtol=1e-5, maxits=100000)
    s.generateBoxMesh(kind, *box, bc_mode=bc)
    s.buildPattern()
    s.assemble(ed, H.TIMEDATA)
    s.profileSpmv(8)
    its, reason, _ = s.factoriseAndSolve()
    its, reason, _ = s.factoriseAndSolve()
    tm = s.timings()
    rowptr, cols, vals = s.getCSR()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        rowptr.astype(np.int32).tofile(os.path.join(d, "rowptr.i32"))
        cols.astype(np.int32).tofile(os.path.join(d, "cols.i32"))
        vals.tofile(os.path.join(d, "vals.f64"))
        s.free()
        r = subprocess.run([os.path.join(ROOT, "tools", "lab", "ilu0_probe"), os.path.join(d, "rowptr.i32"), os.path.join(d, "cols.i32"),
                            os.path.join(d, "vals.f64"), "10"], capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-2000:])
    probe = json.loads(r.stdout.strip().splitlines()[-1])
    spmv_ms = tm["spmv_ms_total"] / max(tm["spmv_launches"], 1) - tm["event_overhead_ms"]
    it_ms = tm["solve_ms"] / max(its, 1)
    out = {"case": what, "free_dofs": sz["size_global"],
           "jacobi_pcg": {"iterations": its, "reason": reason, "solve_ms": tm["solve_ms"], "ms_per_iteration": it_ms, "spmv_ms": spmv_ms},
           "ilu0_apply": probe,
           # an ILU(0)-PCG iteration = this library's iteration with the Jacobi scaling replaced by the two triangular solves
           "ilu0_pcg_ms_per_iteration_estimate": it_ms + probe["lower_solve_ms"] + probe["upper_solve_ms"],
           "iterations_ilu0_may_cost_at_most_for_a_tie": its * it_ms / (it_ms + probe["lower_solve_ms"] + probe["upper_solve_ms"])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
