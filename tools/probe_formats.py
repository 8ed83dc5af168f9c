"""A/B of the SpMV encodings inside the CG on one box (development aid): tools/probe_formats.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H, drivers as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mesh = H.gen_box_tets(-1, 1, n, -1, 1, n, -1, 1, n)
dm, conn, xyz, edof = D._setup(pf.POISSON_TET, mesh)
s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied); s.buildPattern()
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
s.profileSpmv(True)
for rep in range(2):
    for fmt in ("auto", "gaps16", "int32"):
        s.setSpmvFormat(fmt)
        its, reason, rn = s.factoriseAndSolve(); tm = s.timings()
        sp = tm["spmv_ms_total"] / max(tm["spmv_launches"], 1) - tm["event_overhead_ms"]
        print(f"{fmt:7s} rows/lane {s.spmvRowGroup()} bits {s.spmvColumnBits()}: solve {tm['solve_ms']:.1f} ms its {its} "
              f"per-iter {tm['solve_ms']/its*1e3:.1f} us spmv {sp*1e3:.1f} us standalone {s.benchSpmv(30)*1e3:.1f} us")
