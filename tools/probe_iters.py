"""Iteration counts (Jacobi-PCG, rtol 1e-5) of candidate weak-scaling workloads (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pfemfort_amd as pf
from pfemfort_amd import host as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for label, (nx, ny, nz, z1) in {"cube n": (n, n, n, 1.0), "z-stack x2": (n, n, 2 * n, 3.0), "z-stack x4": (n, n, 4 * n, 7.0),
                                "z-stack x8": (n, n, 8 * n, 15.0), "cube 2n (x8)": (2 * n, 2 * n, 2 * n, 1.0),
                                "2x2x2 blocks = cube 2n on [-1,3]^3": (2 * n, 2 * n, 2 * n, 3.0)}.items():
    if label.startswith("2x2x2"):
        mesh = H.gen_box_tets(-1, 3, nx, -1, 3, ny, -1, 3, nz)
    else:
        mesh = H.gen_box_tets(-1, 1, nx, -1, 1, ny, -1, z1, nz)
    r = pf.tetrapoissonparallelimpl1(mesh, rtol=1e-5)
    print(f"{label:40s} {nx}x{ny}x{nz} its {r.its} reason {r.reason}")
