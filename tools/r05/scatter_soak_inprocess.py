#!/usr/bin/env python3
"""Round-4's once-seen wrong scatter matrix, soaked where it is cheap: ONE process, the full-size beam (4.5 M elements,
103 M entries), the atomic-scatter assembly repeated N times against the ORACLE's serial element loop, with the things that
surrounded the original failure in between: multi-GB solvers created and destroyed (pattern builds through the block pool, pool
trims), gamg set-ups on the same solver, the gather form, downloads of the matrix.  Prints one line per block of repetitions
and dumps every differing entry (tests/test_gpu_full_size.py does the same)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import pfemfort_amd as pf
from oracle import pfem_oracle as O
from pfemfort_amd import drivers as D
from pfemfort_amd import host as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
box = (-0.5, 0.5, 50, 0.0, 6.0, 300, -0.5, 0.5, 50, 1, 3)
mesh = H.gen_box_tets(*box[:9], bc_mode=1, ndof=3)
dm, conn, xyz, edof = D._setup(pf.ELAST_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.ELAST_TET, conn, xyz, edof, dm.solnApplied)
s.buildPattern()
om = O.gen_box_tets(*box)
odm = O.dof_numbering(om.nNode, 3, om.bc_node, om.bc_dof, om.bc_val)
oedof = O.elem_dof_array(om.conn, odm.NodeDofArrayNew)
O.set_threads(max(1, min(os.cpu_count() or 1, 64)))
rowptr, cols = O.csr_pattern(oedof, odm.size_global)
O.set_threads(1)
o_vals, o_rhs = O.assemble(O.ELAST_TET, om.xyz, om.conn, oedof, odm.solnApplied, O.ELAST_ELEMDATA, odm.size_global, rowptr, cols)
tol = 1e-12 * np.abs(o_vals).max()
ftol = 1e-12 * np.abs(o_rhs).max()
print(f"beam config 4: {dm.size_global} dofs, {len(o_vals)} entries; oracle ready", flush=True)
bad_runs = 0
worst = 0.0
t0 = time.time()
other = None
for i in range(N):
    # what surrounded the failing run: other multi-GB solvers coming and going through the pool
    if i % 10 == 0:
        if other is not None:
            other.free(); other = None
        else:
            other = pf.PetscSolver().initialise(*[H.box_slab_sizes(120, 120, 120, 0, 1)[k] for k in ("size_local", "size_global")])
            other.generateBoxMesh(pf.POISSON_TET, -1, 1, 120, -1, 1, 120, -1, 1, 120, bc_mode=0)
            other.buildPattern()
            other.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    if i % 25 == 7:
        s.setPreconditioner("gamg"); s.setTolerances(rtol=1e-5, maxits=1000); s.factoriseAndSolve(); s.setPreconditioner("jacobi")
    if i % 15 == 3:
        s.buildPattern()               # a fresh pattern: new value arrays out of the pool
    if i % 4 == 1:
        s.setAssemblyMode("gather"); s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    s.setAssemblyMode("scatter")
    s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    v, f = s.getCSR()[2], s.getRHS()
    dv = np.abs(v - o_vals).max(); df = np.abs(f - o_rhs).max()
    worst = max(worst, dv / tol * 1e-12)
    if not (dv <= tol and df <= ftol):
        bad_runs += 1
        bad = np.nonzero(np.abs(v - o_vals) > tol)[0]
        rows = np.searchsorted(rowptr, bad, side="right") - 1
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        np.savez(os.path.join(out, f"scatter_mismatch_inprocess_{i}.npz"), slot=bad, row=rows, rowlen=np.diff(rowptr)[rows], col=cols[bad], scatter=v[bad], oracle=o_vals[bad])
        print(f"rep {i}: MISMATCH K by {dv:.3e} in {len(bad)} entries, rows {rows[:8].tolist()}, scatter/oracle {(v[bad[:8]] / o_vals[bad[:8]]).tolist()}; F by {df:.3e}", flush=True)
    if (i + 1) % 25 == 0:
        print(f"rep {i + 1}: {bad_runs} mismatching assemblies so far, worst |K - K_oracle| / max|K| {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {N} scatter assemblies of the full-size beam against the oracle, {bad_runs} mismatches, worst relative difference {worst:.2e}")
sys.exit(1 if bad_runs else 0)
