"""Generate tests/golden/elements.npz from the REFERENCE's own Fortran element routines.

Run in the build container only (needs /root/reference + flang): `make -C oracle ref` compiles
elementutilitiesbasisfuncs.F / elementutilitiespoisson.F / elementutilitieselasticity3D.F in
place (the last with the documented 2-token patch of SURVEY finding 5) into
oracle/_ref/libpfem_ref.so; this script calls them on fixed inputs and stores inputs + outputs.
The .npz is data (inputs and expected outputs), not reference source.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pfem_oracle as O  # noqa: E402


def random_tets(n, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.normal(size=(3, 4 * n)) * rng.uniform(0.05, 3.0, size=(3, 1))
    conn = np.arange(4 * n, dtype=np.int32).reshape(n, 4).T.copy()
    a, b, c, d = (xyz[:, conn[i]] for i in range(4))
    jac = np.einsum("ij,ij->j", a - c, np.cross((b - c).T, (d - c).T).T)
    neg = jac < 0
    conn[0, neg], conn[1, neg] = conn[1, neg].copy(), conn[0, neg].copy()
    return xyz, conn


def random_trias(n, seed):
    rng = np.random.default_rng(seed)
    xy = rng.normal(size=(2, 3 * n))
    conn = np.arange(3 * n, dtype=np.int32).reshape(n, 3).T.copy()
    p1, p2, p3 = (xy[:, conn[i]] for i in range(3))
    jac = (p2 - p1)[0] * (p3 - p1)[1] - (p2 - p1)[1] * (p3 - p1)[0]
    neg = jac < 0
    conn[1, neg], conn[2, neg] = conn[2, neg].copy(), conn[1, neg].copy()
    return xy, conn


def main():
    O.build(ref=True)
    assert O.ref_lib() is not None, "oracle/_ref not built (needs /root/reference + flang)"
    g = os.path.join(ROOT, "tests", "golden")
    out = {}
    tet10 = O.read_mesh(os.path.join(g, "input", "tet10"))
    K, F = O.ref_eval_elems(O.POISSON_TET, tet10.xyz, tet10.conn, O.POISSON_ELEMDATA)
    out["tet10_poisson_K"], out["tet10_poisson_F"] = K, F
    sub = tet10.conn[:, ::10].copy()
    K, F = O.ref_eval_elems(O.ELAST_TET, tet10.xyz, sub, O.ELAST_ELEMDATA)
    out["tet10_elast_conn"], out["tet10_elast_K"], out["tet10_elast_F"] = sub, K, F
    xyz, conn = random_tets(300, 20261001)
    aniso = np.array([1.3, 0.7, 2.1])
    K, F = O.ref_eval_elems(O.POISSON_TET, xyz, conn, aniso)
    out.update(rt_xyz=xyz, rt_conn=conn, rt_aniso=aniso, rt_poisson_K=K, rt_poisson_F=F)
    steel = np.array([210.0, 0.25, 1.0, 0.0, -9.81, 0.5])
    K, F = O.ref_eval_elems(O.ELAST_TET, xyz, conn, steel)
    out.update(rt_elast_data=steel, rt_elast_K=K, rt_elast_F=F)
    tria = O.read_mesh(os.path.join(g, "input", "tria20x20"))
    K, F = O.ref_eval_elems(O.POISSON_TRIA, tria.xyz, tria.conn, np.array([1.0, 1.0]))
    out["tria20_K"], out["tria20_F"] = K, F
    xy, c3 = random_trias(300, 7)
    K, F = O.ref_eval_elems(O.POISSON_TRIA, xy, c3, np.array([2.5, 0.4]))
    out.update(rtri_xy=xy, rtri_conn=c3, rtri_data=np.array([2.5, 0.4]), rtri_K=K, rtri_F=F)
    cook = O.read_mesh(os.path.join(g, "input", "cookmembranetria32"))
    e2d = np.array([O.F32(240.565), O.F32(0.3), 0.7, 0.3, -1.1])
    K, F = O.ref_eval_elems(O.ELAST_TRIA, cook.xyz, cook.conn, e2d)
    out.update(cook_elast_data=e2d, cook_elast_K=K, cook_elast_F=F)
    # output step: the reference's own writervtk.F on fixed inputs -> golden VTK text (gzip)
    import ctypes as C
    import gzip
    import tempfile
    rng = np.random.default_rng(0)
    pid = (np.arange(tet10.nElem) * 7 // tet10.nElem).astype(np.int32)
    sol = (tet10.xyz ** 2).sum(0) - 1.7 + rng.standard_normal(tet10.nNode) * 1e-3
    sol[5:10] = [-1e-9, 0.0, -0.0, 12345.6789, 0.0000005]
    vec = rng.standard_normal((tet10.nNode, 3))
    out.update(vtk_procid=pid, vtk_scalar=sol, vtk_vector=vec)
    L = O.ref_lib()
    for name, field, ndof in (("tet10_scalar", sol, 1), ("tet10_vector", vec.ravel(), 3)):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "o.vtk").encode()
            coords = np.asfortranarray(tet10.xyz.T)
            c1 = np.asfortranarray((tet10.conn.T + 1).astype(np.int32))
            L.ref_write_vtk(C.c_int(3), C.c_int(tet10.nElem), C.c_int(tet10.nNode), C.c_int(4), C.c_int(ndof),
                            coords.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p),
                            pid.ctypes.data_as(C.c_void_p), np.ascontiguousarray(field).ctypes.data_as(C.c_void_p),
                            path, C.c_int(len(path)))
            with open(path, "rb") as f, gzip.GzipFile(os.path.join(g, f"vtk_{name}.vtk.gz"), "wb", mtime=0) as z:
                z.write(f.read())
    np.savez_compressed(os.path.join(g, "elements.npz"), **out)
    print("wrote", os.path.join(g, "elements.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
