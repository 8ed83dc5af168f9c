"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X.

Bars (BASELINE.json north_star): integer maps and sparsity pattern bit-exact; per-element
Ke/Fe bit-exact (the kernels evaluate in the reference's order without FMA contraction);
assembled K/F within 1e-12 relative (atomic adds reorder the sums); solution within 1e-8 of
the converged oracle solution at rtol 1e-10.
"""
import os
import sys

import numpy as np
import pytest

import pfemfort_amd as pf
from oracle import pfem_oracle as O
from pfemfort_amd import host as H

pytestmark = pytest.mark.gpu

K_RTOL = 1e-12     # assembled matrix / rhs entries, relative to the largest entry
U_ATOL = 1e-8      # solution vs oracle PCG at rtol 1e-10


def _omesh(m):
    return O.Mesh(m.xyz, m.conn, m.bc_node, m.bc_dof, m.bc_val)


def _device_problem(kind, mesh, elemData):
    from pfemfort_amd import drivers as D
    dm, conn_new, xyz_new, edof = D._setup(kind, mesh)
    s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
    s.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
    s.buildPattern()
    s.assemble(elemData, H.TIMEDATA)
    return s, dm


@pytest.fixture(scope="module")
def tet10(golden_dir):
    return H.read_mesh(f"{golden_dir}/input/tet10")


@pytest.fixture(scope="module")
def tria20(golden_dir):
    return H.read_mesh(f"{golden_dir}/input/tria20x20")


@pytest.fixture(scope="module")
def beam():
    # small cousin of config 4: [-.5,.5]x[0,6]x[-.5,.5], clamp y=0, 3 dofs per node
    return H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)


@pytest.mark.parametrize("kind,ed", [(pf.POISSON_TET, H.POISSON_ELEMDATA),
                                     (pf.POISSON_TET, np.array([1.3, 0.7, 2.1])),
                                     (pf.ELAST_TET, H.ELAST_ELEMDATA)])
def test_device_element_matrices_bit_exact(tet10, kind, ed):
    s, _ = _device_problem(kind, tet10, ed)
    K, F = s.evalElems(ed, H.TIMEDATA)
    Ko, Fo = O.eval_elems(kind, tet10.xyz, tet10.conn, ed)
    assert np.array_equal(K, Ko)
    assert np.array_equal(F, Fo)


def test_device_tria_elements_bit_exact(tria20):
    for kind in (pf.POISSON_TRIA, pf.POISSON_TRIA_INLINE):
        ed = np.array([1.0, 1.0])
        s, _ = _device_problem(kind, tria20, ed)
        K, F = s.evalElems(ed, H.TIMEDATA)
        Ko, Fo = O.eval_elems(kind, tria20.xyz, tria20.conn, ed)
        assert np.array_equal(K, Ko) and np.array_equal(F, Fo)


@pytest.mark.parametrize("mode", ["gather", "scatter"])
@pytest.mark.parametrize("name", ["tet10", "tria20", "tria20mod", "beam", "beam_partial"])
def test_assembly_matches_oracle(name, mode, request):
    mesh = request.getfixturevalue(name.split("_")[0].replace("mod", ""))
    kind, ed = {"tet10": (pf.POISSON_TET, H.POISSON_ELEMDATA), "tria20": (pf.POISSON_TRIA_INLINE, np.array([1.0, 1.0, 0.0])),
                "tria20mod": (pf.POISSON_TRIA, np.array([1.5, 0.5, 0.0])),
                "beam": (pf.ELAST_TET, H.ELAST_ELEMDATA), "beam_partial": (pf.ELAST_TET, H.ELAST_ELEMDATA)}[name]
    if name == "beam_partial":     # partially constrained nodes + nonzero Dirichlet values (lifting path)
        rng = np.random.default_rng(2)
        keep = rng.random(len(mesh.bc_node)) < 0.6
        extra = np.arange(40, 60, dtype=np.int32)
        mesh = H.Mesh(mesh.xyz, mesh.conn, np.concatenate([mesh.bc_node[keep], extra]),
                      np.concatenate([mesh.bc_dof[keep], extra % 3]).astype(np.int32),
                      np.concatenate([rng.standard_normal(keep.sum()) * 0.01, rng.standard_normal(20) * 0.01]))
    from pfemfort_amd import drivers as D
    dm, conn_new, xyz_new, edof = D._setup(kind, mesh)
    s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
    s.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
    s.buildPattern()
    s.setAssemblyMode(mode)
    s.assemble(ed, H.TIMEDATA)
    prob = O.setup_problem(kind, _omesh(mesh), elemData=ed)
    rowptr, cols, vals = s.getCSR()
    assert dm.size_global == prob.dm.size_global
    assert np.array_equal(rowptr, prob.rowptr)            # pattern: bit-exact
    assert np.array_equal(cols, prob.cols)
    rhs = s.getRHS()
    if mode == "gather":
        # one writer per row, element contributions added in ascending element order ==
        # the serial reference loop: K and F are bit-identical to the oracle
        assert np.array_equal(vals, prob.vals)
        assert np.array_equal(rhs, prob.rhs)
        s.assemble(ed, H.TIMEDATA)                        # and reproducible
        assert np.array_equal(s.getCSR()[2], vals)
    else:
        scale = np.abs(prob.vals).max()
        assert np.abs(vals - prob.vals).max() <= K_RTOL * scale
        assert np.abs(rhs - prob.rhs).max() <= K_RTOL * max(np.abs(prob.rhs).max(), 1e-300)
    if name == "tet10":
        assert (dm.size_global, len(cols)) == (729, 9097)
    if name == "tria20":
        assert (dm.size_global, len(cols)) == (361, 2377)


def _hub_mesh(npts, ndof):
    """A hub node joined to every triangle of a triangulated sphere: the hub's matrix rows have
    ndof*(npts+1) entries -- the long-row cases of the gather kernels (LDS blocks of 128 / 64 threads,
    read-modify-write fallback, > 255 entries: scatter)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(npts)
    pts = rng.standard_normal((npts, 3))
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    tri = ConvexHull(pts).simplices
    xyz = np.round(np.vstack([pts, [[0.03, -0.02, 0.01]]]).T.copy(), 8)
    conn = np.vstack([tri.T, np.full(len(tri), npts)]).astype(np.int32)
    # orient every tet positively: Jac = det[(n1-n3); (n2-n3); (n4-n3)] (elementutilitiesbasisfuncs.F:493-514),
    # and the reference STOPs on a negative one
    a, b, c, d = (xyz[:, conn[i]] for i in range(4))
    bad = np.einsum("ij,ij->j", np.cross((a - c).T, (b - c).T).T, d - c) < 0
    conn[:2, bad] = conn[:2, bad][::-1]
    bn = np.repeat(np.arange(0, npts, 7, dtype=np.int32), ndof)            # every 7th surface node clamped ...
    bd = np.tile(np.arange(ndof, dtype=np.int32), len(bn) // ndof)
    if ndof == 3:                                                           # ... some only partially, beyond three
        keep = np.ones(len(bn), bool); keep[10::5] = False                  # fully fixed nodes (no rigid-body mode)
        bn, bd = bn[keep], bd[keep]
        first = np.repeat(np.array([1, 2, 3], np.int32), 3)
        bn = np.concatenate([first, bn]); bd = np.concatenate([np.tile(np.arange(3, dtype=np.int32), 3), bd])
    bv = 0.01 * np.random.default_rng(1).standard_normal(len(bn))
    return H.Mesh(np.ascontiguousarray(xyz), np.ascontiguousarray(conn), bn, bd.astype(np.int32), bv)


@pytest.mark.parametrize("npts,kind_name", [(24, "poisson"), (40, "poisson"), (100, "poisson"), (140, "poisson"), (200, "poisson"), (300, "poisson"),
                                            (12, "elast"), (20, "elast"), (35, "elast"), (60, "elast"), (100, "elast"), (125, "elast")])
def test_gather_assembly_long_rows(npts, kind_name):
    from pfemfort_amd import drivers as D
    kind, ed, ndof = (pf.POISSON_TET, H.POISSON_ELEMDATA, 1) if kind_name == "poisson" else (pf.ELAST_TET, H.ELAST_ELEMDATA, 3)
    mesh = _hub_mesh(npts, ndof)
    dm, conn_new, xyz_new, edof = D._setup(kind, mesh)
    s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
    s.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
    s.buildPattern()
    s.assemble(ed, H.TIMEDATA)                            # default mode: gather wherever it applies
    prob = O.setup_problem(kind, _omesh(mesh), elemData=ed)
    rowptr, cols, vals = s.getCSR()
    assert np.array_equal(rowptr, prob.rowptr) and np.array_equal(cols, prob.cols)
    maxlen = int(np.diff(rowptr).max())
    assert maxlen >= ndof * (npts + 1 - len(np.unique(mesh.bc_node)))       # the hub row really is that long
    # The gather form owns every row of up to 255 entries (the byte-sized entry index of its records; the LDS row
    # accumulators take up to 160 KiB per workgroup): bit-exact, one writer.  A node with a longer row is a HUB: its rows
    # alone go through a scatter pass with atomics (sums reordered), the rest of the mesh is untouched by that.
    info = s.assemblyInfo()
    lens = np.diff(rowptr)
    hub_rows = np.nonzero(lens > 255)[0]
    assert info["gather"] and info["hub_nodes"] == len(hub_rows) // ndof and (len(hub_rows) > 0) == (maxlen > 255)
    keep = np.ones(len(vals), bool)
    for r in hub_rows:
        keep[rowptr[r]:rowptr[r + 1]] = False
    rhs = s.getRHS()
    assert np.array_equal(vals[keep], prob.vals[keep])
    assert np.array_equal(np.delete(rhs, hub_rows), np.delete(prob.rhs, hub_rows))
    assert np.abs(vals - prob.vals).max() <= K_RTOL * np.abs(prob.vals).max()
    assert np.abs(rhs - prob.rhs).max() <= K_RTOL * max(np.abs(prob.rhs).max(), 1e-300)
    if ndof == 3 and 128 < maxlen <= 255:
        assert info["lds_bytes"] > 65536                 # rows this long need the large LDS allocation of gfx950
    its, reason, _ = s.factoriseAndSolve()
    assert reason > 0


def test_spmv_matches_oracle(tet10):
    s, dm = _device_problem(pf.POISSON_TET, tet10, H.POISSON_ELEMDATA)
    rowptr, cols, vals = s.getCSR()
    rng = np.random.default_rng(7)
    x = rng.standard_normal(dm.size_global)
    y = s.spmv(x)
    yo = O.spmv(rowptr, cols, vals, x)
    assert np.abs(y - yo).max() <= 1e-13 * np.abs(yo).max()


@pytest.mark.parametrize("name", ["tet10", "beam", "beam_partial", "tria20", "cook"])
def test_spmv_column_formats_bit_identical(name, request):
    """Row-grouped / 16-bit column gaps / int32 columns: same products, same order, same bits."""
    mesh = request.getfixturevalue(name.split("_")[0])
    kind, ed = {"tet10": (pf.POISSON_TET, H.POISSON_ELEMDATA), "beam": (pf.ELAST_TET, H.ELAST_ELEMDATA),
                "beam_partial": (pf.ELAST_TET, H.ELAST_ELEMDATA), "cook": (pf.ELAST_TRIA, H.ELAST2D_ELEMDATA),
                "tria20": (pf.POISSON_TRIA, np.array([1.0, 1.0]))}[name]
    if name == "beam_partial":     # nodes with one or two free dofs: groups of 1, 2 and 3 rows
        rng = np.random.default_rng(2)
        keep = rng.random(len(mesh.bc_node)) < 0.6
        extra = np.arange(40, 70, dtype=np.int32)
        mesh = H.Mesh(mesh.xyz, mesh.conn, np.concatenate([mesh.bc_node[keep], extra]),
                      np.concatenate([mesh.bc_dof[keep], extra % 3]).astype(np.int32), np.zeros(keep.sum() + 30))
    s, dm = _device_problem(kind, mesh, ed)
    x = np.random.default_rng(1).standard_normal(dm.size_global)
    assert s.spmvRowGroup() == 1               # "auto" keeps small systems in the row form
    s.setSpmvFormat("grouped")
    y_auto = s.spmv(x)
    # beam: the 3 dof rows of a node share a lane; structured scalar meshes: 4 consecutive rows share a relative
    # column stream; Cook's membrane (2 dofs per node, unstructured) stays in the row form
    assert s.spmvRowGroup() == {"beam": 3, "beam_partial": 3, "tet10": 4, "tria20": 1, "cook": 1}[name]
    s.setSpmvFormat("gaps16")
    assert s.spmvRowGroup() == 1 and np.array_equal(y_auto, s.spmv(x))
    s.setSpmvFormat("int32")
    y_32 = s.spmv(x)
    assert np.array_equal(y_auto, y_32)
    rowptr, cols, vals = s.getCSR()
    assert np.abs(y_32 - O.spmv(rowptr, cols, vals, x)).max() <= 1e-13 * np.abs(y_32).max()


@pytest.mark.parametrize("name", ["cube", "beam"])
def test_spmv_value_dictionary_same_bits(name, monkeypatch):
    """The SpMV's group forms stream 16-bit codes into a dictionary of the DISTINCT matrix values when there are at most 4096 of
    them (pfem_valdict.hpp: structured meshes repeat their element matrices).  Lossless: product, Jacobi-CG history and iterate,
    gamg history and iterate equal the fp64 copy's bit for bit (PFEM_SPMV_VALDICT=0); new values of the same pattern are
    re-encoded (a value missing from the old dictionary rebuilds it); a mesh whose values do not repeat keeps the fp64 copy."""
    if name == "cube":
        kind, ed, ed2 = pf.POISSON_TET, H.POISSON_ELEMDATA, np.array([1.3, 0.7, 2.1])
        mesh = H.gen_box_tets(-1, 1, 24, -1, 1, 20, -1, 1, 22)
        h = 2.0 / 24
    else:
        kind, ed = pf.ELAST_TET, H.ELAST_ELEMDATA
        ed2 = np.array(H.ELAST_ELEMDATA, dtype=np.float64).copy()
        ed2[0] *= 1.7                       # another Young's modulus: every entry changes
        mesh = H.gen_box_tets(-0.5, 0.5, 6, 0.0, 6.0, 36, -0.5, 0.5, 6, bc_mode=1, ndof=3)
        h = 1.0 / 6
    out = {}
    for vd in ("0", "1"):
        monkeypatch.setenv("PFEM_SPMV_VALDICT", vd)
        s, dm = _device_problem(kind, mesh, ed)
        s.setSpmvFormat("grouped")
        s.buildPattern()
        s.assemble(ed, H.TIMEDATA)
        x = np.random.default_rng(3).standard_normal(dm.size_global)
        y = s.spmv(x)
        n_dict = s.spmvValueDictionary()
        assert (n_dict > 0) == (vd == "1") and n_dict <= 4096 and s.spmvRowGroup() == (4 if name == "cube" else 3)
        s.setTolerances(rtol=1e-10, maxits=20000)
        its, reason, _ = s.factoriseAndSolve()
        hj, uj = s.getHistory(), s.getSolution()
        s.setPreconditioner("gamg")
        itg, reason_g, _ = s.factoriseAndSolve()
        hg, ug = s.getHistory(), s.getSolution()
        assert reason == 2 and reason_g == 2
        # other values on the same pattern: the codes follow (the old dictionary misses them, a new one is collected)
        s.assemble(ed2, H.TIMEDATA)
        y2 = s.spmv(x)
        n2 = s.spmvValueDictionary()
        assert (n2 > 0) == (vd == "1") and not np.array_equal(y, y2)
        s.assemble(ed, H.TIMEDATA)
        y3 = s.spmv(x)
        assert np.array_equal(y3, y)
        out[vd] = (y, its, hj, uj, itg, hg, ug, y2)
        bytes_now = s.spmvFormatBytes()
        out[vd + "bytes"] = bytes_now
        s.free()
    a, b = out["0"], out["1"]
    assert a[1] == b[1] and a[4] == b[4]
    for i in (0, 2, 3, 5, 6, 7):
        assert np.array_equal(a[i], b[i]), i
    assert out["1bytes"] < 0.5 * out["0bytes"]                     # 2 B a slot instead of 8
    # nodes moved off the lattice: no two element matrices alike, the dictionary overflows and the fp64 copy stays
    monkeypatch.setenv("PFEM_SPMV_VALDICT", "1")
    s, dm = _device_problem(kind, _moved(mesh, h), ed)
    s.setSpmvFormat("grouped")
    s.buildPattern()
    s.assemble(ed, H.TIMEDATA)
    x = np.random.default_rng(3).standard_normal(dm.size_global)
    y = s.spmv(x)
    assert s.spmvValueDictionary() == 0
    rowptr, cols, vals = s.getCSR()
    assert np.abs(y - O.spmv(rowptr, cols, vals, x)).max() <= 1e-13 * np.abs(y).max()
    s.assemble(ed, H.TIMEDATA)
    assert np.array_equal(s.spmv(x), y) and s.spmvValueDictionary() == 0          # (and it is not tried again on this pattern)


@pytest.mark.parametrize("rtol", [1e-5, 1e-10])
def test_poisson_tet10_solve(tet10, rtol):
    res = pf.tetrapoissonparallelimpl1(tet10, rtol=rtol)
    prob = O.setup_problem(O.POISSON_TET, _omesh(tet10))
    x, its, reason, rn, hist = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=rtol, hist_len=200)
    assert res.reason == reason == 2
    assert abs(res.its - its) <= 1
    h = res.solver.getHistory()
    n = min(len(h), len(hist))
    assert np.allclose(h[:n], hist[:n], rtol=1e-6)
    xc, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    tol = U_ATOL if rtol <= 1e-10 else 1e-3
    assert np.abs(res.soln_free - xc).max() <= tol
    if rtol <= 1e-10:
        exact = (tet10.xyz ** 2).sum(0)                    # u = x^2+y^2+z^2, nodally exact
        assert np.abs(res.solnVTK[:, 0] - exact).max() < 2e-7


def test_compat_driver_equals_batched(tet10):
    a = pf.tetrapoissonparallelimpl1(tet10, rtol=1e-10, mode="batched")
    b = pf.tetrapoissonparallelimpl1(tet10, rtol=1e-10, mode="compat")
    ra, ca, va = a.solver.getCSR()
    rb, cb, vb = b.solver.getCSR()
    assert np.array_equal(ra, rb) and np.array_equal(ca, cb)
    prob = O.setup_problem(O.POISSON_TET, _omesh(tet10))
    assert np.array_equal(vb, prob.vals)                   # host-staged serial sums: bit-exact
    assert np.array_equal(b.solver.getRHS(), prob.rhs)
    assert np.abs(va - vb).max() <= K_RTOL * np.abs(vb).max()
    assert np.abs(a.soln_free - b.soln_free).max() <= U_ATOL


def test_tria20x20_known_answer(tria20):
    res = pf.triapoissonserialimpl1(tria20, rtol=1e-12)
    assert res.reason > 0
    assert abs(res.soln_free.sum() - 68.09843993245326) < 1e-8       # SURVEY 8c (direct solve)
    assert np.allclose(res.soln_free[:3], [0.13364425, 0.26399773, 0.38785071], atol=1e-7)
    x, y = tria20.xyz
    exact = np.sin(np.pi * x) * (np.cosh(np.pi * y) - np.cosh(np.pi) / np.sinh(np.pi) * np.sinh(np.pi * y))
    assert abs(np.abs(res.solnVTK[:, 0] - exact).max() - 7.11e-4) < 1e-5


def test_elasticity_beam_solve(beam):
    res = pf.tetraelasticityparallelimpl1(beam, rtol=1e-10)
    prob = O.setup_problem(O.ELAST_TET, _omesh(beam))
    xc, its, reason, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-10)
    assert res.reason == reason == 2 and abs(res.its - its) <= 2
    assert np.abs(res.soln_free - xc).max() <= U_ATOL * max(1.0, np.abs(xc).max())
    assert res.solnVTK[:, 0].max() > 0.1                    # the beam bends in +x under (0.1,0,0)


@pytest.mark.parametrize("partial", [False, True])
def test_node_block_jacobi_matches_oracle(beam, partial):
    """-pc_type pbjacobi.  Same blocks as the oracle's rule (rows with identical column sets),
    same convergence test; fewer iterations than point Jacobi, same solution."""
    mesh = beam
    if partial:       # nodes with one or two free dofs -> blocks of 1, 2 and 3 rows
        rng = np.random.default_rng(2)
        keep = rng.random(len(mesh.bc_node)) < 0.8
        extra = np.arange(40, 70, dtype=np.int32)
        mesh = H.Mesh(mesh.xyz, mesh.conn, np.concatenate([mesh.bc_node[keep], extra]),
                      np.concatenate([mesh.bc_dof[keep], extra % 3]).astype(np.int32), np.zeros(keep.sum() + 30))
    s, dm = _device_problem(pf.ELAST_TET, mesh, H.ELAST_ELEMDATA)
    prob = O.setup_problem(O.ELAST_TET, _omesh(mesh))
    s.setTolerances(rtol=1e-10, maxits=20000)
    its_j, reason_j, _ = s.factoriseAndSolve()
    u_j = s.getSolution()
    assert s.preconditioner() == "jacobi"
    s.setPreconditioner("pbjacobi")
    assert s.preconditioner() == "pbjacobi"
    its_b, reason_b, rn_b = s.factoriseAndSolve()
    u_b = s.getSolution()
    groups = O.row_groups(prob.rowptr, prob.cols)
    x, its_o, reason_o, rn_o = O.pcg_block_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, groups, rtol=1e-10, maxits=20000)
    assert reason_j == reason_b == reason_o == 2
    assert abs(its_b - its_o) <= 2 and its_b < its_j
    assert np.abs(u_b - x).max() <= U_ATOL * max(1.0, np.abs(x).max())
    assert np.abs(u_b - u_j).max() <= U_ATOL * max(1.0, np.abs(x).max())
    if partial:
        assert np.diff(groups).min() < 3 and np.diff(groups).max() == 3
    # Poisson has no multi-row groups: the request falls back to point Jacobi, and says so
    sp, _ = _device_problem(pf.POISSON_TET, H.gen_box_tets(-1, 1, 4, -1, 1, 4, -1, 1, 4), H.POISSON_ELEMDATA)
    sp.setPreconditioner("pbjacobi")
    assert sp.preconditioner() == "jacobi"


def test_negative_jacobian_is_reported(tet10):
    bad = H.Mesh(tet10.xyz, tet10.conn.copy(), tet10.bc_node, tet10.bc_dof, tet10.bc_val)
    bad.conn[[0, 1], 5] = bad.conn[[1, 0], 5]              # flip one tet
    with pytest.raises(pf.PfemError) as ei:
        pf.tetrapoissonparallelimpl1(bad)
    assert ei.value.code == 3                               # PFEM_ERR_NEG_JAC (reference: STOP)


def test_status_machine(tet10):
    s = pf.PetscSolver().initialise(10, 10)
    assert s.currentStatus == pf.solver.SOLVER_EMPTY
    with pytest.raises(pf.PfemError):
        s.factorise()                                       # "Assemble matrix first..." solverpetsc.F:415
    with pytest.raises(pf.PfemError):
        s.solve()


def test_assemble_matrix_and_vector_methods(tet10):
    """PetscSolver%assembleMatrixAndVector / assembleMatrix / assembleVector (solverpetsc.F:328-401):
    per-entry MatSetValue(R(ii), C(jj), K(ii,jj)) -- K is NOT transposed on this route."""
    from pfemfort_amd import drivers as D
    dm, conn, xyz, edof = D._setup(pf.POISSON_TET, tet10)
    N = dm.size_global
    s = pf.PetscSolver().initialise(N, N)
    z16 = np.zeros(16)
    for e in range(tet10.nElem):
        s.MatSetValues(edof[:, e], edof[:, e], z16, pf.solver.INSERT_VALUES)
    s.setZero()
    rng = np.random.default_rng(3)
    A = np.zeros((N, N)); b = np.zeros(N)
    for e in range(0, tet10.nElem, 7):                       # any values: this checks the plumbing
        K = rng.standard_normal((4, 4)); F = rng.standard_normal(4)
        f = edof[:, e]
        if e % 14 == 0:
            s.assembleMatrixAndVector(f, f, K, F)
        else:
            s.assembleMatrix(f, f, K)
            s.assembleVector(f, F)
        ok = f >= 0
        A[np.ix_(f[ok], f[ok])] += K[np.ix_(ok, ok)]
        b[f[ok]] += F[ok]
    s.setTolerances(maxits=0)
    s.factoriseAndSolve()                                    # pushes the staged values to the device
    rowptr, cols, vals = s.getCSR()
    dense = np.zeros((N, N))
    for r in range(N):
        dense[r, cols[rowptr[r]:rowptr[r + 1]]] = vals[rowptr[r]:rowptr[r + 1]]
    assert np.allclose(dense, A, rtol=0, atol=1e-14) and np.allclose(s.getRHS(), b, rtol=0, atol=1e-14)


@pytest.fixture(scope="module")
def cook(golden_dir):
    return H.read_mesh(f"{golden_dir}/input/cookmembranetria32")


@pytest.mark.parametrize("mode", ["gather", "scatter"])
def test_elast_tria_cook_membrane(cook, mode):
    """2-D sibling behind the same boundary: plane-stress triangles, nodal forces."""
    from pfemfort_amd import drivers as D
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    ed = np.array([H.ELAST2D_ELEMDATA[0], H.ELAST2D_ELEMDATA[1], 0.7, 0.3, -1.1])
    dm, conn_new, xyz_new, edof = D._setup(pf.ELAST_TRIA, cook)
    s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
    s.uploadMesh(pf.ELAST_TRIA, conn_new, xyz_new, edof, dm.solnApplied)
    s.buildPattern()
    s.setAssemblyMode(mode)
    K, F = s.evalElems(ed, H.TIMEDATA)
    Ko, Fo = O.eval_elems(O.ELAST_TRIA, cook.xyz, cook.conn, ed)
    assert np.array_equal(K, Ko) and np.array_equal(F, Fo)          # device element matrices: bit-exact
    s.assemble(ed, H.TIMEDATA)
    prob = O.setup_problem(O.ELAST_TRIA, _omesh(cook), elemData=ed)
    rowptr, cols, vals = s.getCSR()
    assert np.array_equal(rowptr, prob.rowptr) and np.array_equal(cols, prob.cols)
    if mode == "gather":
        assert np.array_equal(vals, prob.vals) and np.array_equal(s.getRHS(), prob.rhs)
    else:
        assert np.abs(vals - prob.vals).max() <= K_RTOL * np.abs(prob.vals).max()
    # the driver with its intended semantics: unit thickness, no body force, ForceBC nodal loads
    res = pf.triaelasticityparallelimpl1(cook, rtol=1e-12, maxits=100000)
    prob = O.setup_problem(O.ELAST_TRIA, _omesh(cook))
    rhs = prob.rhs.copy()
    rhs[prob.dm.NodeDofArrayNew[cook.force_node, cook.force_dof]] += cook.force_val
    u = spl.spsolve(sp.csr_matrix((prob.vals, prob.cols, prob.rowptr)).tocsc(), rhs)
    assert res.reason == 2 and np.abs(res.soln_free - u).max() <= 1e-8 * np.abs(u).max()
    tip = res.solnVTK[np.argmax(cook.xyz[0] + cook.xyz[1])]          # corner (48, 60)
    assert tip[1] > 0.0                                               # the membrane is sheared upwards


def test_tria_poisson_parallel_driver(tria20):
    a = pf.triapoissonparallelimpl1(tria20, rtol=1e-12)
    b = pf.triapoissonserialimpl1(tria20, rtol=1e-12)                 # inline element: same problem to rounding
    assert np.abs(a.soln_free - b.soln_free).max() < 1e-10


def test_edge_cases_of_the_solver_boundary(tet10):
    """Empty / degenerate / out-of-contract inputs behave like PETSc's documented semantics."""
    # (1) zero right-hand side: converged at iteration 0 on the absolute tolerance (KSP_CONVERGED_ATOL)
    s = pf.PetscSolver().initialise(3, 3)
    idx = np.arange(3, dtype=np.int32)
    s.MatSetValues(idx, idx, np.zeros(9), pf.solver.INSERT_VALUES)
    s.setZero()
    s.MatSetValues(idx, idx, np.array([2.0, -1, 0, -1, 2, -1, 0, -1, 2]), pf.solver.ADD_VALUES)
    assert s.factoriseAndSolve()[:2] == (0, 3) and np.array_equal(s.getSolution(), np.zeros(3))
    # (2) negative indices are ignored in rows, columns and the rhs; ragged m != n blocks are fine
    s.setZero()
    s.MatSetValues([0, -1, 2], [1, -1], np.array([5.0, 9, 9, 9, 7, 9]), pf.solver.ADD_VALUES)
    s.MatSetValues(idx, idx, np.diag([4.0, 4, 4]).ravel(), pf.solver.ADD_VALUES)
    s.VecSetValues([-1, 1, 2], [100.0, 4.0, 8.0], pf.solver.ADD_VALUES)
    s.setTolerances(rtol=1e-14)
    its, reason, _ = s.factoriseAndSolve()
    A = np.array([[4.0, 5, 0], [0, 4, 0], [0, 7, 4]])
    assert reason in (2, -10, -8) or its >= 0          # a non-symmetric toy matrix: only the plumbing is asserted
    rowptr, cols, vals = s.getCSR()
    dense = np.zeros((3, 3))
    for r in range(3):
        dense[r, cols[rowptr[r]:rowptr[r + 1]]] = vals[rowptr[r]:rowptr[r + 1]]
    assert np.array_equal(dense, A) and np.array_equal(s.getRHS(), [0.0, 4.0, 8.0])
    # (3) ADD_VALUES outside the inserted pattern is an error (MAT_NEW_NONZERO_LOCATIONS stays off after setZero)
    t = pf.PetscSolver().initialise(3, 3)
    t.MatSetValues([0], [0], np.zeros(1), pf.solver.INSERT_VALUES)
    t.setZero()
    with pytest.raises(pf.PfemError) as ei:
        t.MatSetValues([0], [2], np.ones(1), pf.solver.ADD_VALUES)
    assert ei.value.code == 8
    # (4) iteration limit: KSP_DIVERGED_ITS, solution is the last iterate
    r = pf.tetrapoissonparallelimpl1(tet10, rtol=1e-14, maxits=5)
    assert (r.its, r.reason) == (5, -3)
    # (5) a single free dof (everything else Dirichlet)
    free = 665                                                        # an interior node of tet10
    keep = np.ones(tet10.nNode, bool); keep[free] = False
    exact = (tet10.xyz ** 2).sum(0)
    m1 = H.Mesh(tet10.xyz, tet10.conn, np.nonzero(keep)[0].astype(np.int32), np.zeros(keep.sum(), np.int32), exact[keep])
    r1 = pf.tetrapoissonparallelimpl1(m1, rtol=1e-12)
    assert r1.dm.size_global == 1 and r1.reason > 0 and abs(r1.soln_free[0] - exact[free]) < 1e-10
    # (6) mesh with zero elements on this rank is legal (an idle rank): pattern of 0 entries
    z = pf.PetscSolver().initialise(0, 0)
    z.uploadMesh(pf.POISSON_TET, np.empty((4, 0), np.int32), tet10.xyz, np.empty((4, 0), np.int32), np.zeros(tet10.nNode))
    z.buildPattern()
    z.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    assert z.matrixInfo()["nnz"] == 0 and z.factoriseAndSolve()[1] == 3


def test_indefinite_matrix_is_reported_like_petsc():
    """(p, A p) <= 0 stops the solve with KSP_DIVERGED_INDEFINITE_MAT (-10 in PETSc's enum), same as the oracle loop."""
    n = 50
    s = pf.PetscSolver().initialise(n, n)
    idx = np.arange(n, dtype=np.int32)
    for i in range(n):
        c = [j for j in (i - 1, i, i + 1) if 0 <= j < n]
        s.MatSetValues([i], c, np.zeros(len(c)), pf.solver.INSERT_VALUES)
    s.setZero()
    for i in range(n):                       # -(1-D Laplacian) with a positive diagonal preconditioner is not possible:
        c = [j for j in (i - 1, i, i + 1) if 0 <= j < n]          # diag +2, off-diagonals +3 -> indefinite, diag > 0
        s.MatSetValues([i], c, np.array([2.0 if j == i else 3.0 for j in c]), pf.solver.ADD_VALUES)
    s.VecSetValues(idx, (-1.0) ** np.arange(n), pf.solver.ADD_VALUES)      # p0 = r0/2 alternates: (p0, A p0) < 0
    its, reason, _ = s.factoriseAndSolve()
    rowptr, cols, vals = s.getCSR()
    _, its_o, reason_o, *_ = O.pcg_jacobi(rowptr, cols, vals, s.getRHS(), rtol=1e-5)
    assert reason == reason_o == -10 and its == its_o == 1


def test_blocks_that_start_late_still_apply_the_last_step(monkeypatch):
    """The direction kernel publishes the verdict of an iteration while other blocks of the SAME launch may not have
    started yet (low occupancy, shared device, profiler).  With 60 kB of dynamic LDS per block only two blocks fit a
    CU, so most of the 2048 blocks start after the lead block is done: the solution must not change by a bit."""
    mesh = H.gen_box_tets(-1, 1, 100, -1, 1, 100, -1, 1, 100)
    a = pf.tetrapoissonparallelimpl1(mesh, rtol=1e-5)
    monkeypatch.setenv("PFEM_DEBUG_DIRECTION_LDS", "60000")
    b = pf.tetrapoissonparallelimpl1(mesh, rtol=1e-5)
    assert (a.its, a.reason) == (b.its, b.reason) and a.reason == 2
    assert np.array_equal(a.soln_free, b.soln_free)


@pytest.mark.parametrize("kind,box,bc_mode,nparts", [(pf.POISSON_TET, (6, 5, 7), 0, 1), (pf.POISSON_TET, (6, 5, 7), 0, 3),
                                                     (pf.ELAST_TET, (3, 6, 5), 1, 1), (pf.ELAST_TET, (3, 6, 5), 1, 2)])
def test_device_generated_box_equals_host_generator_and_bookkeeping(kind, box, bc_mode, nparts):
    """pfem_mesh_generate_box (mesh + numbering evaluated on the device, one z-slab per part) against the host path --
    pfem_gen_box_tets, pfem_partition_box_slabs, pfem_dof_numbering, pfem_renumber_mesh, pfem_elem_dof_array,
    pfem_mesh_upload -- array by array, bit for bit; then the assembled K, F."""
    nEx, nEy, nEz = box
    ndof = 3 if kind == pf.ELAST_TET else 1
    ext = (-1.0, 1.0, nEx, -0.5, 1.5, nEy, 0.0, 3.0, nEz)
    ed = H.ELAST_ELEMDATA if kind == pf.ELAST_TET else H.POISSON_ELEMDATA
    plane = (nEx + 1) * (nEy + 1)
    for part in range(nparts):
        sz = H.box_slab_sizes(nEx, nEy, nEz, bc_mode, ndof, nparts, part)
        k0, k1 = nEz * part // nparts, nEz * (part + 1) // nparts
        a = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"], row_start=sz["row_start"])
        a.generateBoxMesh(kind, *ext, bc_mode=bc_mode, nparts=nparts, part=part)
        conn_a, xyz_a, edof_a, sa_a = a.downloadMesh()
        # host path of the same slab (nodes of the whole grid, global ids)
        mesh = H.gen_box_tets(*ext, bc_mode=bc_mode, ndof=ndof, kz=(k0, k1))
        _, npid = H.partition_box_slabs(nEx, nEy, nEz, nparts, elements=False)
        dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, nparts, npid)
        conn_new, xyz_new = H.renumber_mesh(mesh, dm)
        edof = H.elem_dof_array(conn_new, dm.NodeDofArrayNew)
        b = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"], row_start=sz["row_start"])
        b.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
        _, _, edof_b, _ = b.downloadMesh()
        assert np.array_equal(conn_a + plane * k0, conn_new)
        assert np.array_equal(xyz_a, xyz_new[:, plane * k0:plane * (k1 + 1)])
        assert np.array_equal(sa_a, dm.solnApplied[plane * k0 * ndof:plane * (k1 + 1) * ndof])
        assert np.array_equal(edof_a, edof_b)
        assert np.array_equal(a.ghosts(), b.ghosts())
        for s in (a, b):
            s.buildPattern()
            s.assemble(ed, H.TIMEDATA)
        for x, y in zip(a.getCSR(), b.getCSR()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.getRHS(), b.getRHS())


@pytest.mark.parametrize("kind,box,bc_mode,nparts,axis", [(pf.POISSON_TET, (6, 5, 7), 0, 3, 0), (pf.POISSON_TET, (6, 5, 7), 0, 2, 1),
                                                          (pf.ELAST_TET, (3, 7, 4), 1, 3, 1), (pf.ELAST_TET, (5, 3, 4), 1, 2, 0),
                                                          (pf.ELAST_TET, (3, 7, 4), 1, 3, -1), (pf.POISSON_TET, (4, 4, 4), 0, 2, -1)])
def test_device_generated_box_along_any_axis_equals_host_path(kind, box, bc_mode, nparts, axis):
    """pfem_mesh_generate_box_axis: slabs of hex layers across x or y (BASELINE config 4's beam is cut across its length)
    renumber the mesh like the reference does for any partition (tetrapoissonparallelimpl1.F:541-612: ranks concatenated,
    ascending old id inside a rank) -- no longer the identity.  The device evaluates that numbering in closed form;
    here it is compared with the host path driven by the same node_proc_id (pfem_partition_box_slabs_axis,
    pfem_dof_numbering, pfem_renumber_mesh, pfem_elem_dof_array, pfem_mesh_upload): connectivity, coordinates,
    prescribed values through the node map, the element dof array, the ghosts, then pattern, K and F -- bit for bit."""
    nEx, nEy, nEz = box
    ndof = 3 if kind == pf.ELAST_TET else 1
    ext = (-1.0, 1.0, nEx, -0.5, 1.5, nEy, 0.0, 3.0, nEz)
    ed = H.ELAST_ELEMDATA if kind == pf.ELAST_TET else H.POISSON_ELEMDATA
    mesh = H.gen_box_tets(*ext, bc_mode=bc_mode, ndof=ndof)
    epid, npid = H.partition_box_slabs(nEx, nEy, nEz, nparts, axis=axis)
    dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, nparts, npid)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)
    used = None
    for part in range(nparts):
        sz = H.box_slab_sizes(nEx, nEy, nEz, bc_mode, ndof, nparts, part, axis=axis)
        used = sz["axis"]
        assert used == (axis if axis >= 0 else (1 if box == (3, 7, 4) else 2))            # the longest axis, ties to z
        a = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"], row_start=sz["row_start"])
        a.generateBoxMesh(kind, *ext, bc_mode=bc_mode, nparts=nparts, part=part, axis=axis)
        conn_a, xyz_a, edof_a, sa_a = a.downloadMesh()
        # local node (x-fastest over the slab's node box) -> old id -> new id
        l0, l1 = sz["layer0"], sz["layer1"]
        rng = [np.arange(nEx + 1), np.arange(nEy + 1), np.arange(nEz + 1)]
        rng[used] = np.arange(l0, l1 + 1)
        kk, jj, ii = np.meshgrid(rng[2], rng[1], rng[0], indexing="ij")
        old = ((kk * (nEy + 1) + jj) * (nEx + 1) + ii).ravel()
        new = dm.node_map_get_new[old]
        mine = np.nonzero(epid == part)[0]
        conn_loc = np.ascontiguousarray(conn_new[:, mine])
        assert np.array_equal(new[conn_a], conn_loc)
        assert np.array_equal(xyz_a, xyz_new[:, new])
        # prescribed values: the reference re-enters them at the new ids WITHOUT clearing the old slots (:668-677, SURVEY
        # A.2 step 4 -- only Dirichlet-typed slots are ever read), the device never writes those stale entries
        dirichlet = dm.NodeDofArrayNew[new] < 0
        assert np.array_equal(sa_a.reshape(-1, ndof)[dirichlet], dm.solnApplied.reshape(-1, ndof)[new][dirichlet])
        assert not sa_a.reshape(-1, ndof)[~dirichlet].any()
        edof = H.elem_dof_array(conn_loc, dm.NodeDofArrayNew)
        b = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"], row_start=sz["row_start"])
        b.uploadMesh(kind, conn_loc, xyz_new, edof, dm.solnApplied)
        _, _, edof_b, _ = b.downloadMesh()
        assert np.array_equal(edof_a, edof_b)
        assert np.array_equal(a.ghosts(), b.ghosts())
        assert (sz["row_start"], sz["row_start"] + sz["size_local"]) == (dm.row_start[part], dm.row_end[part])
        for s in (a, b):
            s.buildPattern()
            s.assemble(ed, H.TIMEDATA)
        for x, y in zip(a.getCSR(), b.getCSR()):
            assert np.array_equal(x, y)
        assert np.array_equal(a.getRHS(), b.getRHS())
        assert np.array_equal(a.localToGlobal(), b.localToGlobal())


@pytest.mark.parametrize("form", ["dictionary", "gap32"])
def test_relative_row_groups_beyond_16bit_gaps(form, monkeypatch):
    """A plane of more than 65 535 free nodes (here 257^2; BASELINE config 5's slabs have 399^2): the z-neighbour is further
    away than a 16-bit gap can say.  The relative-row-group SpMV keeps 16-bit codes with a table of the few distinct
    large gaps (k_spmvr<., true>) or -- forced here, and whenever the table overflows -- streams one 32-bit gap per entry
    (k_spmvr32) instead of falling back to int32 columns per row: same products, same order, same bits; and the solve
    agrees with the oracle."""
    if form == "gap32":
        monkeypatch.setenv("PFEM_DEBUG_REL_GAP32", "1")
    mesh = H.gen_box_tets(-1, 1, 258, -1, 1, 258, -1, 1, 4)
    s, dm = _device_problem(pf.POISSON_TET, mesh, H.POISSON_ELEMDATA)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(dm.size_global)
    s.setSpmvFormat("int32")
    y32 = s.spmv(x)
    assert (s.spmvRowGroup(), s.spmvColumnBits()) == (1, 32)
    s.setSpmvFormat("grouped")                                   # the group forms whatever the size
    assert (s.spmvRowGroup(), s.spmvColumnBits()) == (4, 16 if form == "dictionary" else 32)
    assert (1 <= s.spmvGapTable() <= 16) if form == "dictionary" else s.spmvGapTable() == 0    # a handful of plane-sized gaps
    assert np.array_equal(s.spmv(x), y32)
    nnz = s.matrixInfo()["nnz"]
    assert s.spmvFormatBytes() < (0.76 if form == "dictionary" else 0.8) * (12 * nnz + 20 * dm.size_global)
    s.setTolerances(rtol=1e-10)
    its, reason, _ = s.factoriseAndSolve()
    prob = O.setup_problem(O.POISSON_TET, _omesh(mesh))
    xo, its_o, reason_o, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-10)
    assert reason == reason_o == 2 and abs(its - its_o) <= 1
    assert np.abs(s.getSolution() - xo).max() <= U_ATOL


def test_relative_row_groups_with_many_distinct_large_gaps():
    """More distinct gaps >= 32768 than the 256-entry table holds (a band that wanders: 489 different far offsets, each
    constant over 512 rows): the table overflows and the 32-bit gap stream is taken; y equals the int32 row form bit for
    bit.  Through MatSetValues, so the compat path's patterns reach the group forms too."""
    n = 250000
    s = pf.PetscSolver().initialise(n, n)
    rng = np.random.default_rng(3)
    far = 66000 + 97 * ((np.arange(n) // 512) % 600)           # 489 blocks of 512 rows: 489 distinct gaps beyond 65535
    I, A = pf.solver.INSERT_VALUES, pf.solver.ADD_VALUES
    rows_cols = []
    for i in range(n):
        c = [j for j in (i - 1, i, i + 1, i + int(far[i])) if 0 <= j < n]
        rows_cols.append(np.array(c, np.int32))
        s.MatSetValues([i], rows_cols[-1], np.zeros(len(c)), I)
    s.setZero()
    for i in range(n):
        s.MatSetValues([i], rows_cols[i], rng.standard_normal(len(rows_cols[i])), A)
    s.VecSetValues(np.arange(8, dtype=np.int32), np.ones(8), A)
    s.setTolerances(rtol=1e-5, maxits=1)
    s.factoriseAndSolve()                      # pushes the staged matrix to the device (the random matrix is not SPD:
    x = rng.standard_normal(n)                 # whatever the verdict of that one iteration, only the SpMV is compared)
    s.setSpmvFormat("int32")
    y32 = s.spmv(x)
    s.setSpmvFormat("grouped")
    assert (s.spmvRowGroup(), s.spmvColumnBits()) == (4, 32)
    assert np.array_equal(s.spmv(x), y32)
    rowptr, cols, vals = s.getCSR()
    assert np.abs(y32 - O.spmv(rowptr, cols, vals, x)).max() <= 1e-13 * np.abs(y32).max()


@pytest.mark.parametrize("name", ["tet10", "tria20", "beam", "cube40"])
def test_single_reduction_cg_matches_its_oracle(name, request, monkeypatch):
    """KSPCGUseSingleReduction (pfem_solver_set_cg_single_reduction / PFEM_CG_SINGLE_REDUCTION): the device loop against
    the oracle's restatement of the same recurrences -- iteration count +-1, residual history, reasons -- and against
    the converged two-reduction solution."""
    monkeypatch.setenv("PFEM_CG_SINGLE_REDUCTION", "1")
    if name == "cube40":                    # large enough for the relative row groups and a folded (p,Ap)
        mesh = H.gen_box_tets(-1, 1, 40, -1, 1, 40, -1, 1, 40)
        run, kind, kw = pf.tetrapoissonparallelimpl1, O.POISSON_TET, {}
    else:
        mesh = request.getfixturevalue(name)
        run, kind, kw = {"tet10": (pf.tetrapoissonparallelimpl1, O.POISSON_TET, {}),
                         "tria20": (pf.triapoissonserialimpl1, O.POISSON_TRIA_INLINE, {"elemData": np.array([1.0, 1.0, 0.0])}),
                         "beam": (pf.tetraelasticityparallelimpl1, O.ELAST_TET, {})}[name]
    prob = O.setup_problem(kind, _omesh(mesh), **kw)
    for rtol in (1e-5, 1e-10):
        res = run(mesh, rtol=rtol)
        x, its, reason, rn, hist = O.pcg_jacobi_single_reduction(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=rtol, hist_len=4000)
        assert res.reason == reason == 2 and abs(res.its - its) <= (3 if name == "beam" else 1)
        h = res.solver.getHistory()
        n = min(len(h), len(hist))
        assert n >= min(its, res.its)
        if name != "beam":                   # (the slender beam amplifies rounding differences along the history)
            assert np.allclose(h[:n], hist[:n], rtol=1e-6)
        if rtol == 1e-10:
            xc, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
            assert np.abs(res.soln_free - xc).max() <= U_ATOL * max(1.0, np.abs(xc).max())
    # iteration limit, as the two-reduction loop reports it
    r = run(mesh, rtol=1e-14, maxits=5)
    assert (r.its, r.reason) == (5, -3)
    x5, its5, reason5, *_ = O.pcg_jacobi_single_reduction(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-14, maxits=5)
    assert np.allclose(r.soln_free, x5, rtol=1e-10, atol=1e-13)


def test_single_reduction_cg_edge_reasons():
    """Zero right-hand side (converged at iteration 0), an indefinite matrix (-10 at iteration 1) and the setter
    overriding the environment, in the single-reduction form."""
    n = 50
    s = pf.PetscSolver().initialise(n, n)
    s.setSingleReduction(True)
    idx = np.arange(n, dtype=np.int32)
    for i in range(n):
        c = [j for j in (i - 1, i, i + 1) if 0 <= j < n]
        s.MatSetValues([i], c, np.zeros(len(c)), pf.solver.INSERT_VALUES)
    s.setZero()
    for i in range(n):
        c = [j for j in (i - 1, i, i + 1) if 0 <= j < n]
        s.MatSetValues([i], c, np.array([2.0 if j == i else 3.0 for j in c]), pf.solver.ADD_VALUES)
    assert s.factoriseAndSolve()[:2] == (0, 3)                        # b = 0
    s.VecSetValues(idx, (-1.0) ** np.arange(n), pf.solver.ADD_VALUES)
    its, reason, _ = s.factoriseAndSolve()
    rowptr, cols, vals = s.getCSR()
    _, its_o, reason_o, *_ = O.pcg_jacobi_single_reduction(rowptr, cols, vals, s.getRHS(), rtol=1e-5)
    assert reason == reason_o == -10 and its == its_o == 1
    s.setTolerances(rtol=1e-5, maxits=0)
    assert s.factoriseAndSolve()[:2] == (0, -3)


def test_gather_kernel_variants_agree_bit_for_bit(monkeypatch):
    """The tet-Poisson gather assembly ships in two kernels (32-B node records + XCD-contiguous chunk order; the generic
    structure-of-arrays kernel the 2-D kinds also use) and two block orders: the same additions in the same order, so
    the same bits -- and the oracle's literal serial loop, which keeps the reference's products by 0 and 1."""
    mesh = H.gen_box_tets(-1, 1, 33, -1, 1, 29, -1, 1, 31)           # 31 k nodes: > 64 blocks, so the XCD order is on
    # some coordinates become -0.0 (what "-0.00000000" in a node file parses to): the lean geometry's only difference
    xyz = mesh.xyz.copy()
    xyz[xyz == 0.0] = -0.0
    rng = np.random.default_rng(4)
    extra = rng.choice(mesh.nNode, 300, replace=False).astype(np.int32)   # interior Dirichlet nodes: lifting inside the box
    mesh = H.Mesh(xyz, mesh.conn, np.concatenate([mesh.bc_node, extra]), np.zeros(len(mesh.bc_node) + 300, np.int32),
                  np.concatenate([mesh.bc_val, rng.standard_normal(300)]))
    from pfemfort_amd import drivers as D
    dm, conn_new, xyz_new, edof = D._setup(pf.POISSON_TET, mesh)
    prob = O.setup_problem(O.POISSON_TET, _omesh(mesh))
    got = {}
    for name, env in (("node4_xcd", {}), ("node4_plain", {"PFEM_DEBUG_GATHER_PLAIN_ORDER": "1"}), ("soa", {"PFEM_DEBUG_GATHER_SOA": "1"})):
        for k in ("PFEM_DEBUG_GATHER_PLAIN_ORDER", "PFEM_DEBUG_GATHER_SOA"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
        s.uploadMesh(pf.POISSON_TET, conn_new, xyz_new, edof, dm.solnApplied)
        s.buildPattern()
        s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
        got[name] = (s.getCSR()[2], s.getRHS())
        s.free()
    for name, (vals, rhs) in got.items():
        assert np.array_equal(vals.view(np.uint64), prob.vals.view(np.uint64)), name      # bits, zero signs included
        assert np.array_equal(rhs.view(np.uint64), prob.rhs.view(np.uint64)), name


@pytest.mark.parametrize("name,per_range", [("tet10", 1000), ("tet10", 6000), ("beam", 37)])
def test_pattern_built_in_element_ranges(name, per_range, request, monkeypatch):
    """Meshes with more than 2^31-1 element-matrix entries per device (config 5 alone on one MI355X: 6.1e9) get their
    pattern from element ranges, each sorted and made unique on its own (pfem_pattern_build).  Forced here on small
    meshes: pattern, K and F equal the oracle's bit for bit, as on the one-pass path."""
    monkeypatch.setenv("PFEM_DEBUG_PATTERN_RANGE", str(per_range))
    mesh = request.getfixturevalue(name)
    kind, ed = (pf.POISSON_TET, H.POISSON_ELEMDATA) if name == "tet10" else (pf.ELAST_TET, H.ELAST_ELEMDATA)
    from pfemfort_amd import drivers as D
    dm, conn_new, xyz_new, edof = D._setup(kind, mesh)
    s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
    s.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
    s.buildPattern()
    s.assemble(ed, H.TIMEDATA)
    prob = O.setup_problem(kind, _omesh(mesh), elemData=ed)
    rowptr, cols, vals = s.getCSR()
    assert np.array_equal(rowptr, prob.rowptr) and np.array_equal(cols, prob.cols)
    assert np.array_equal(vals, prob.vals) and np.array_equal(s.getRHS(), prob.rhs)
    its, reason, _ = s.factoriseAndSolve()
    assert reason == 2


@pytest.mark.parametrize("name", ["tet10", "beam", "tria20"])
def test_pattern_from_incidence_lists_equals_the_sorted_keys(name, request, monkeypatch):
    """The default pattern build collects every node's distinct neighbour nodes from the incidence lists (k_pattern_rows);
    PFEM_DEBUG_PATTERN_SORT=1 sorts the keys of every element-matrix entry instead (the fallback, and the form of patterns
    without a mesh).  Same rowptr, same columns, and both equal the oracle's."""
    mesh = request.getfixturevalue(name)
    kind, ed = {"tet10": (pf.POISSON_TET, H.POISSON_ELEMDATA), "beam": (pf.ELAST_TET, H.ELAST_ELEMDATA),
                "tria20": (pf.POISSON_TRIA, H.POISSON_ELEMDATA)}[name]
    from pfemfort_amd import drivers as D
    dm, conn_new, xyz_new, edof = D._setup(kind, mesh)
    prob = O.setup_problem(kind, _omesh(mesh), elemData=ed)
    got = {}
    for form in ("lists", "sorted"):
        if form == "sorted":
            monkeypatch.setenv("PFEM_DEBUG_PATTERN_SORT", "1")
        s = pf.PetscSolver().initialise(dm.size_global, dm.size_global)
        s.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
        s.buildPattern()
        s.assemble(ed, H.TIMEDATA)
        got[form] = s.getCSR() + (s.getRHS(),)
        s.free()
    for form, (rowptr, cols, vals, rhs) in got.items():
        assert np.array_equal(rowptr, prob.rowptr) and np.array_equal(cols, prob.cols), form
        assert np.array_equal(vals, prob.vals) and np.array_equal(rhs, prob.rhs), form


@pytest.mark.parametrize("table", [True, False, "escapes"])
def test_row_forms_beyond_16bit_gaps(table, monkeypatch):
    """Three dofs per node and a numbering plane of 151 x 151 nodes: the gap to the next plane's columns is 68 403 dofs,
    more than a literal 16-bit gap holds.  The row form and the 3-row form keep their 16-bit streams through the table of
    distinct large gaps (k_spmv16 / k_spmvg <., true>); without it (whenever the table overflows) the row form escapes the large
    gaps to the int32 column array (k_spmv16e; "escapes"), and without that either (round 1) the solver drops to int32 columns per row.  Same products, same order: bit-identical y; K, F equal the oracle's; the solve agrees."""
    escapes = table == "escapes"         # no table, but gaps beyond 16 bits may escape to the int32 column (k_spmv16e): 16-bit row form again
    table = table is True
    if not table:
        monkeypatch.setenv("PFEM_DEBUG_NO_ROW_GAP_TABLE", "1")
        if not escapes:
            monkeypatch.setenv("PFEM_DEBUG_NO_GAP_ESCAPES", "1")
    mesh = H.gen_box_tets(-0.5, 0.5, 150, 0.0, 1.0, 150, -0.01, 0.01, 1, bc_mode=1, ndof=3)
    s, dm = _device_problem(pf.ELAST_TET, mesh, H.ELAST_ELEMDATA)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(dm.size_global)
    s.setSpmvFormat("int32")
    y32 = s.spmv(x)
    assert (s.spmvRowGroup(), s.spmvColumnBits(), s.spmvGapTable()) == (1, 32, 0)
    s.setSpmvFormat("grouped")
    if table:
        assert (s.spmvRowGroup(), s.spmvColumnBits()) == (3, 16) and 1 <= s.spmvGapTable() <= 64
    else:
        assert s.spmvRowGroup() != 3          # no 3-row form without 16-bit row streams (the relative groups may step in)
    assert np.array_equal(s.spmv(x), y32)
    s.setSpmvFormat("gaps16")                                   # one row per lane
    assert (s.spmvRowGroup(), s.spmvColumnBits()) == ((1, 16) if (table or escapes) else (1, 32))
    assert np.array_equal(s.spmv(x), y32)
    assert s.spmvGapEscapes() == bool(escapes)
    if escapes:
        assert s.spmvGapTable() == 0
    if table:
        nnz = s.matrixInfo()["nnz"]
        s.setSpmvFormat("grouped")                              # ("auto" keeps a system this small in the row form)
        assert s.spmvRowGroup() == 3 and s.spmvFormatBytes() < 0.78 * (12 * nnz + 20 * dm.size_global)
        prob = O.setup_problem(O.ELAST_TET, _omesh(mesh))
        rowptr, cols, vals = s.getCSR()
        assert np.array_equal(cols, prob.cols) and np.array_equal(vals, prob.vals) and np.array_equal(s.getRHS(), prob.rhs)
        # (a thin clamped plate: Jacobi-PCG would need ~1e5 iterations -- the first 60 iterates are compared instead)
        s.setTolerances(rtol=1e-30, maxits=60)
        its, reason, _ = s.factoriseAndSolve()
        xo, its_o, reason_o, rn_o, hist_o = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-30, maxits=60, hist_len=61)
        assert (its, reason) == (its_o, reason_o) == (60, -3)
        assert np.abs(s.getSolution() - xo).max() <= 1e-9 * np.abs(xo).max()
        assert np.allclose(s.getHistory()[:61], hist_o, rtol=1e-8)


# ---------------------------------------------------------------------------------------------------------------
# -pc_type gamg: plain-aggregation multigrid as the preconditioner of the CG (pfem_amg.inc)
# ---------------------------------------------------------------------------------------------------------------
def _transfers(s, info=None):
    """What the oracle's cycle is given, level by level: the aggregates the device formed (coarse dof of every dof) or, where the
    transfer carries rigid-body modes, the oracle's OWN prolongator built from the node aggregates and the node coordinates
    (level 0: the mesh nodes; below: the centroids the oracle computed itself, compared with the device's)."""
    info = info or s.amgInfo()
    out, cen = [], None
    for l in range(info["levels"] - 1):
        a = s.amgAggregates(l, info["rows"][l])
        tr = s.amgTransfer(l)
        if not tr["rbm"]:
            out.append(a)
            cen = None
            continue
        fb, cb, dim = tr["fine_bs"], tr["coarse_bs"], tr["dim"]
        assert cb == dim + (3 if dim == 3 else 1) and fb in (dim, cb) and info["rows"][l] == fb * tr["n_nodes"]
        a2 = a.reshape(-1, fb)
        assert not (a2[:, 0] % cb).any() and all(np.array_equal(a2[:, c], a2[:, 0] + c) for c in range(fb))     # translation part of P
        dev_xyz = s.amgTransfer(l, xyz=True)["xyz"]
        if cen is None:
            cen = dev_xyz
        assert np.abs(cen - dev_xyz).max() <= 1e-12 * max(1.0, np.abs(dev_xyz).max())
        P, cen = O.rbm_prolongator(a2[:, 0] // cb, cen, dim, fb)
        out.append(P)
    return out


def _gamg_vs_oracle(s, rtol=1e-10):
    """Solve with gamg on the device, then restate the SAME solve on the CPU (oracle.pcg_amg) with the aggregates the
    device formed: iteration count, residual history and solution."""
    s.setTolerances(rtol=rtol, maxits=10000)
    s.setPreconditioner("gamg")
    its, reason, rn = s.factoriseAndSolve()
    assert s.preconditioner() == "gamg"
    info = s.amgInfo()
    aggs = _transfers(s, info)
    rowptr, cols, vals = s.getCSR()
    cyc = s.amgCycle()          # V on lattice bricks, W on matched aggregates (or what the test asked for)
    xo, ito, ro, rno, hist = O.pcg_amg(rowptr, cols, vals, s.getRHS(), aggs, cheb_degree=info["cheb_degree"], fine_degree=info["fine_degree"], eig_ratio=info["eig_ratio"],
                                       coarse_scale=info["coarse_scale"], rtol=rtol, gamma=2 if cyc["cycle"] == "w" else 1,
                                       gamma_to=cyc["last_level_visited_twice"])
    x = s.getSolution()
    h = s.getHistory()
    assert (reason, ro) == (2, 2) and abs(its - ito) <= max(1, ito // 50), (its, ito)      # (a run of hundreds of iterations: +-2 %)
    m = min(len(h), len(hist), 30)       # rounding differences grow along a long CG run (Cook's membrane: 100+ iterations)
    assert np.abs(h[:m] - hist[:m]).max() <= 1e-6 * hist[0]
    assert np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())
    return its, info, aggs, x


def _moved(mesh, h, frac=0.2, seed=7):
    """The mesh with every node that carries no Dirichlet value moved inside a ball of radius frac * h (h = the smallest cell
    edge; Kuhn's tetrahedra keep their orientation up to 0.28): the coordinates no longer form a lattice."""
    d = np.random.default_rng(seed).uniform(-1.0, 1.0, size=mesh.xyz.shape) * (frac * h / np.sqrt(3.0))
    d[:, np.unique(mesh.bc_node)] = 0.0
    return H.Mesh(mesh.xyz + d, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val)


@pytest.mark.parametrize("case", ["tet10", "cube30", "beam", "cook", "compat", "tria20", "tiny", "aniso", "cube30_moved", "beam_moved",
                                  "cube30_moved_w", "beam_moved_w", "tet10_w"])
def test_gamg_solve_equals_oracle_restatement(case, tet10, beam, tria20, golden_dir, monkeypatch):
    """-pc_type gamg on file meshes and generated boxes, scalar and 3-dof problems, the batched and the MatSetValues path:
    the device hierarchy (matching aggregates, Galerkin sums, Gershgorin bounds, Chebyshev V-cycle, dense bottom solve) and
    its PCG loop against the oracle's restatement given the same aggregates; against a direct solve; fewer iterations than
    point Jacobi; aggregates are what three passes of pairing can produce."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    # (the moved boxes keep a box's numbering, which would give them their bricks back -- test_gamg_lattice_by_numbering...; here they
    # stand for meshes without any lattice: matching on the strength graph)
    monkeypatch.setenv("PFEM_AMG_LATTICE_BY_NUMBERING", "0")
    if case == "compat":
        res = pf.tetrapoissonparallelimpl1(tet10, mode="compat", rtol=1e-10)
        s, its_j = res.solver, res.its
    else:
        kind, mesh, ed = {"tet10": (pf.POISSON_TET, tet10, H.POISSON_ELEMDATA),
                          "cube30": (pf.POISSON_TET, H.gen_box_tets(-1, 1, 30, -1, 1, 30, -1, 1, 30), H.POISSON_ELEMDATA),
                          "beam": (pf.ELAST_TET, beam, H.ELAST_ELEMDATA),
                          # conductivity 100x larger along z: pairs across the weak axes are refused (couplings below a quarter of the
                          # strongest) and an axis that pairs next to nothing is passed over: semi-coarsening along z, on the lattice
                          "aniso": (pf.POISSON_TET, H.gen_box_tets(-1, 1, 20, -1, 1, 20, -1, 1, 20), np.array([1.0, 1.0, 100.0])),
                          "tria20": (pf.POISSON_TRIA_INLINE, tria20, None),                         # config 1's mesh: 361 dofs, two levels
                          "tiny": (pf.POISSON_TET, H.gen_box_tets(-1, 1, 4, -1, 1, 4, -1, 1, 4), H.POISSON_ELEMDATA),   # 27 dofs: no coarse level at all
                          # nodes moved off the lattice: no positions, the aggregates come from matching on the strength graph;
                          # _w: -pc_mg_cycle_type w (every coarse problem above the cycle's tail visited twice) against the oracle's W
                          "cube30_moved": (pf.POISSON_TET, _moved(H.gen_box_tets(-1, 1, 30, -1, 1, 30, -1, 1, 30), 2.0 / 30), H.POISSON_ELEMDATA),
                          "cube30_moved_w": (pf.POISSON_TET, _moved(H.gen_box_tets(-1, 1, 30, -1, 1, 30, -1, 1, 30), 2.0 / 30), H.POISSON_ELEMDATA),
                          "tet10_w": (pf.POISSON_TET, tet10, H.POISSON_ELEMDATA),
                          "beam_moved": (pf.ELAST_TET, _moved(H.gen_box_tets(-0.5, 0.5, 6, 0.0, 6.0, 36, -0.5, 0.5, 6, bc_mode=1, ndof=3), 1.0 / 6), H.ELAST_ELEMDATA),
                          "beam_moved_w": (pf.ELAST_TET, _moved(H.gen_box_tets(-0.5, 0.5, 6, 0.0, 6.0, 36, -0.5, 0.5, 6, bc_mode=1, ndof=3), 1.0 / 6), H.ELAST_ELEMDATA),
                          "cook": (pf.ELAST_TRIA, H.read_mesh(f"{golden_dir}/input/cookmembranetria32"), H.ELAST2D_ELEMDATA)}[case]
        s, dm = _device_problem(kind, mesh, ed)
        if case.endswith("_w"):
            s.setAmgCycle("w")
        if case in ("beam", "beam_moved", "beam_moved_w"):
            s.setSpmvFormat("grouped")          # the 3-row node groups (and with them the node-wise aggregation) at this size too
            s.buildPattern()
            s.assemble(ed, H.TIMEDATA)
        if case == "cook" and mesh.force_node is not None:
            s.addNodalForces(dm.NodeDofArrayNew[dm.node_map_get_new[mesh.force_node], mesh.force_dof], mesh.force_val)
        s.setTolerances(rtol=1e-10, maxits=20000)
        its_j, reason_j, _ = s.factoriseAndSolve()
        assert reason_j == 2
    its, info, aggs, x = _gamg_vs_oracle(s)
    assert its < its_j or case == "tiny", (its, its_j)
    rowptr, cols, vals = s.getCSR()
    u = spl.spsolve(sp.csr_matrix((vals, cols, rowptr)).tocsc(), s.getRHS())
    assert np.abs(x - u).max() <= 1e-8 * max(1.0, np.abs(u).max())
    # hierarchy: every level at least 1.25x smaller, the last one small enough for the dense inverse; aggregates of at most
    # 8 nodes (x 3 dofs each on the beam, whose dofs stay with their node)
    rows = info["rows"]
    if case == "tiny":                          # fewer rows than the dense bottom takes: the cycle is the Chebyshev polynomial alone
        assert info["levels"] == 1 and its <= its_j
        return
    assert info["levels"] >= 2 and all(10 * b <= 8 * a for a, b in zip(rows, rows[1:])) and rows[-1] <= 128
    rbm = [s.amgTransfer(l)["rbm"] for l in range(info["levels"] - 1)]
    assert all(rbm) if case in ("beam", "cook", "beam_moved", "beam_moved_w") else not any(rbm)       # displacement problems: rigid-body modes on every level
    cyc = s.amgCycle()
    assert cyc["cycle"] == ("w" if case.endswith("_w") else "v")
    if cyc["cycle"] == "w":         # (a hierarchy of two levels has no coarse problem to visit twice: its W is its V)
        assert (1 <= cyc["last_level_visited_twice"] <= info["levels"] - 2) or info["levels"] <= 2
    for l, (a, n_c) in enumerate(zip(aggs, rows[1:])):
        if rbm[l]:        # nodes per aggregate (every coarse node has dim translations + rotations)
            tr = s.amgTransfer(l)
            cnt = np.bincount(s.amgAggregates(l, rows[l]).reshape(-1, tr["fine_bs"])[:, 0] // tr["coarse_bs"])
            # (level 0 of a displacement problem: five passes of pairing, 32 nodes -- on a lattice, where the odd node of a line joins
            # its neighbour's pair, up to 4 x 6 x 3; below: three passes, 8 nodes, on a lattice up to 3 x 3 x 3)
            # (no lattice: roots + neighbours -- a root, everybody around it, and the leftovers between: compact, no fixed size)
            kind_l = s.amgAggregation()[l]
            assert len(cnt) * tr["coarse_bs"] == n_c and cnt.min() >= 1 and cnt.max() <= (96 if kind_l == "roots" else (72 if l == 0 else (27 if case == "beam" else 8)))
            continue
        cnt = np.bincount(a)
        # (three passes of pairing: at most 8; on the beam's lattice the node a line of odd length leaves over joins the pair next
        # to it: bricks of up to 3 along every axis)
        assert len(cnt) == n_c and cnt.min() >= 1 and cnt.max() <= 8
    lat = s.amgLayout()["lattice_levels"]       # generated boxes and the reference's tet10 file sit on a lattice, Cook's membrane does not
    assert (lat >= 1) if case in ("tet10", "cube30", "beam", "tria20", "aniso", "tet10_w") else (lat == 0 or case == "compat")
    if case == "aniso":       # the weak axes are passed over: the passes of the first level pair along z (19 -> 10 -> 5 -> 3 per column,
        assert 900 <= rows[1] <= 19 * 19 * 3 and np.bincount(aggs[0]).max() <= 8 and lat >= 2      # a few pairs across y at the end)
        return
    if case == "cube30":                        # a lattice numbered line by line: mostly 2x2x2 bricks
        assert (np.bincount(aggs[0]) == 8).mean() > 0.7 and its <= 0.3 * its_j
    assert all(1.0 < lam < 8.0 for lam in info["lambda_max"])


def _waved(mesh, h, box, amp=0.3):
    """The mesh under a smooth map that is no tensor product (a wavy block: every node moved along all three axes by amp * h times a
    product of sines of the OTHER coordinates, vanishing on the boundary): no lattice of coordinates, the box's numbering intact."""
    lo = np.array([box[0], box[2], box[4]])[:, None]
    ext = np.array([box[1] - box[0], box[3] - box[2], box[5] - box[4]])[:, None]
    u = (mesh.xyz - lo) / ext                                   # unit cube
    bubble = np.prod(np.sin(np.pi * u), axis=0)
    d = np.stack([np.sin(2 * np.pi * u[1]) * np.cos(np.pi * u[2]), np.sin(3 * np.pi * u[2]) * np.cos(np.pi * u[0]),
                  np.sin(2 * np.pi * u[0]) * np.cos(2 * np.pi * u[1])]) * bubble * (amp * h)
    d[:, np.unique(mesh.bc_node)] = 0.0
    return H.Mesh(mesh.xyz + d, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val)


@pytest.mark.parametrize("kind_name", ["poisson", "elast", "poisson_waved"])
def test_gamg_lattice_by_numbering_when_the_nodes_left_their_sites(kind_name, monkeypatch):
    """A box whose nodes were moved (a mapped block, a mesh after a moving-mesh step) has no lattice of coordinates, but its
    NUMBERING is still a box's: the incidence lists are translated copies of a few patterns, their offsets give the strides
    (lattice_positions_by_numbering), every element is held to them, and the hierarchy takes the bricks the unmoved box would
    take -- the oracle's restatement from the UNMOVED coordinates equals the device's aggregates entry for entry; the cycle on
    the moved mesh's matrix equals the oracle's given those aggregates.  The same mesh under a random numbering has neither
    lattice and is left to the passes / matching."""
    monkeypatch.delenv("PFEM_AMG_LATTICE_BY_NUMBERING", raising=False)
    if kind_name.startswith("poisson"):
        kind, ed = pf.POISSON_TET, H.POISSON_ELEMDATA
        mesh0 = H.gen_box_tets(-1, 1, 26, -1, 1, 26, -1, 1, 26)
        h = 2.0 / 26
    else:
        kind, ed = pf.ELAST_TET, H.ELAST_ELEMDATA
        mesh0 = H.gen_box_tets(-0.5, 0.5, 8, 0.0, 6.0, 48, -0.5, 0.5, 8, bc_mode=1, ndof=3)
        h = 1.0 / 8
    # (random displacements inside a ball of 0.2 h; or -- "waved" -- a smooth map that is no tensor product)
    mesh = _waved(mesh0, h, (-1, 1, -1, 1, -1, 1)) if kind_name.endswith("waved") else _moved(mesh0, h)
    assert len(np.unique(np.round(mesh.xyz[0], 12))) > 1024       # (no lattice of coordinates: more distinct values than the table takes)
    s, dm = _device_problem(kind, mesh, ed)
    if kind_name == "elast":
        s.setSpmvFormat("grouped")
        s.buildPattern()
        s.assemble(ed, H.TIMEDATA)
    assert s.incidencePatterns()[0] >= 1
    s.setTolerances(rtol=1e-10, maxits=20000)
    its, info, aggs, x = _gamg_vs_oracle(s)
    kinds = s.amgAggregation()
    assert s.amgLayout()["lattice_levels"] >= 1 and kinds[0] == ("bricks" if kind_name.startswith("poisson") else "node-bricks"), kinds
    xyz0 = mesh0.xyz[:, dm.node_map_get_old]
    if kind_name.startswith("poisson"):
        free = np.where(dm.NodeDofArrayNew.reshape(-1) >= 0)[0]
        own = O.lattice_brick_aggregates(xyz0, xyz0[:, free])
        dev = aggs
    else:
        free = np.where(dm.NodeDofArrayNew.reshape(-1, 3)[:, 0] >= 0)[0]
        own = O.lattice_node_brick_aggregates(xyz0, xyz0[:, free])
        dev = []
        for l in range(info["levels"] - 1):
            tr = s.amgTransfer(l)
            dev.append(s.amgAggregates(l, info["rows"][l]).reshape(-1, tr["fine_bs"])[:, 0] // tr["coarse_bs"])
    assert own is not None and len(own) == len(dev) >= 2 and all(np.array_equal(a, b) for a, b in zip(own, dev)), (kinds, info["rows"])
    s.free()
    s2, _ = _device_problem(kind, _shuffled(mesh), ed)
    s2.setPreconditioner("gamg")
    s2.setTolerances(rtol=1e-10, maxits=20000)
    its2, reason2, _ = s2.factoriseAndSolve()
    assert reason2 == 2 and "bricks" not in s2.amgAggregation() and "node-bricks" not in s2.amgAggregation()
    assert its <= its2
    s2.free()


@pytest.mark.parametrize("case", ["tet10", "cube30", "tria20", "odd"])
def test_gamg_lattice_bricks_equal_the_oracles_own(case, tet10, tria20):
    """On a lattice with strong couplings along every axis the aggregates are bricks of positions.  The oracle restates them
    from the node COORDINATES alone (O.lattice_brick_aggregates: no device data), level by level down the hierarchy: the
    device's aggregate maps are equal to them entry for entry, so the oracle's cycle on these cases is fed nothing the device
    made -- and its solve still matches the device's iteration for iteration."""
    kind, mesh, ed = {"tet10": (pf.POISSON_TET, tet10, H.POISSON_ELEMDATA),
                      "cube30": (pf.POISSON_TET, H.gen_box_tets(-1, 1, 30, -1, 1, 30, -1, 1, 30), H.POISSON_ELEMDATA),
                      "tria20": (pf.POISSON_TRIA_INLINE, tria20, None),
                      "odd": (pf.POISSON_TET, H.gen_box_tets(0, 2.3, 23, 0, 1.7, 17, 0, 1.2, 12), H.POISSON_ELEMDATA)}[case]      # cubic cells, lines of 22, 16, 11 free nodes
    s, dm = _device_problem(kind, mesh, ed)
    s.setPreconditioner("gamg")
    s.setTolerances(rtol=1e-10, maxits=10000)
    its, reason, _ = s.factoriseAndSolve()
    info = s.amgInfo()
    dev = [s.amgAggregates(l, info["rows"][l]) for l in range(info["levels"] - 1)]
    free = np.where(dm.NodeDofArrayNew.reshape(-1) >= 0)[0]
    xyz_new = mesh.xyz[:, dm.node_map_get_old]
    own = O.lattice_brick_aggregates(xyz_new, xyz_new[:, free])
    assert len(own) == len(dev) >= 1 and all(np.array_equal(a, b) for a, b in zip(own, dev))
    assert s.amgLayout()["lattice_levels"] == len(dev)
    rowptr, cols, vals = s.getCSR()
    xo, ito, ro, _, hist = O.pcg_amg(rowptr, cols, vals, s.getRHS(), own, cheb_degree=info["cheb_degree"], fine_degree=info["fine_degree"],
                                     eig_ratio=info["eig_ratio"], coarse_scale=info["coarse_scale"], rtol=1e-10)
    assert (reason, ro) == (2, 2) and abs(its - ito) <= 1 and np.abs(s.getSolution() - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("box", [(8, 48, 8, 0.5), (6, 36, 6, 0.5), (10, 30, 6, 0.3)])
def test_gamg_node_bricks_equal_the_oracles_own(box):
    """Displacement problems on a lattice whose lines are full take their NODE aggregates as bricks in one step (amg_node_bricks:
    3 or 4 nodes to the brick edge on level 0, pairs below, the short brick at the end of a line joined to its neighbour).  The
    oracle restates them from the node coordinates alone (O.lattice_node_brick_aggregates) and the device's node aggregates equal
    them entry for entry on every level that carries the rigid-body transfer."""
    nx, ny, nz, hz = box                    # (cubic cells: a brick needs every coupling inside it to be strong)
    mesh = H.gen_box_tets(-0.5, 0.5, nx, 0.0, ny / nx, ny, -hz, hz, nz, bc_mode=1, ndof=3)
    s, dm = _device_problem(pf.ELAST_TET, mesh, H.ELAST_ELEMDATA)
    s.setSpmvFormat("grouped")
    s.buildPattern()
    s.assemble(H.ELAST_ELEMDATA, H.TIMEDATA)
    s.setPreconditioner("gamg")
    s.setTolerances(rtol=1e-8, maxits=2000)
    its, reason, _ = s.factoriseAndSolve()
    assert reason == 2
    info = s.amgInfo()
    nda = dm.NodeDofArrayNew.reshape(-1, 3)
    free = np.where(nda[:, 0] >= 0)[0]
    xyz_new = mesh.xyz[:, dm.node_map_get_old]
    own = O.lattice_node_brick_aggregates(xyz_new, xyz_new[:, free])
    assert own is not None and len(own) == info["levels"] - 1 >= 2, (None if own is None else [len(a) for a in own], info["rows"])
    for l, a in enumerate(own):
        tr = s.amgTransfer(l)
        assert tr["rbm"]
        dev = s.amgAggregates(l, info["rows"][l]).reshape(-1, tr["fine_bs"])[:, 0] // tr["coarse_bs"]
        assert np.array_equal(dev, a), (l, np.nonzero(dev != a)[0][:10])


_BRICK_VARIANT = r"""
import sys, json, hashlib
import numpy as np
sys.path.insert(0, {root!r})
import pfemfort_amd as pf
from pfemfort_amd import host as H
n = 24
sz = H.box_slab_sizes(n, n, n)
s = pf.PetscSolver().initialise(sz["size_local"], sz["size_global"])
s.generateBoxMesh(pf.POISSON_TET, -1.0, 1.0, n, -1.0, 1.0, n, -1.0, 1.0, n)
s.buildPattern()
s.setPreconditioner("gamg")
s.setTolerances(rtol=1e-10, maxits=1000)
s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
its, reason, rnorm = s.factoriseAndSolve()
info = s.amgInfo()
x = s.getSolution()
print(json.dumps(dict(its=its, reason=reason, rnorm=rnorm, rows=info["rows"], nnz=info["nnz"], lam=info["lambda_max"],
                      x=hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(), xs=float(np.abs(x).sum()))))
"""


def test_gamg_brick_level_variants_agree():
    """A brick level's coarse pattern comes from two walks of every coarse row's member rows and its Galerkin product is formed
    by coarse row (k_lat_codes_*, k_lat_galerkin).  The forms they replaced stay in the library -- the entry -> slot lists
    (PFEM_AMG_GALERKIN_MAPS=1: the same additions in the same order, so the same bits) and the sorted keys
    (PFEM_AMG_BRICK_SORT=1: rows too long for the walks' 16-bit counters; the same hierarchy, sums in slot order) -- and are
    run here, each in a process of its own (the switches are read once)."""
    import subprocess, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, extra in (("default", {}), ("maps", {"PFEM_AMG_GALERKIN_MAPS": "1"}), ("sorted", {"PFEM_AMG_BRICK_SORT": "1"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _BRICK_VARIANT.format(root=root)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d, m, q = out["default"], out["maps"], out["sorted"]
    assert d["reason"] == 2 and d["its"] < 40 and len(d["rows"]) >= 3
    assert (m["its"], m["rnorm"], m["x"], m["lam"], m["rows"], m["nnz"]) == (d["its"], d["rnorm"], d["x"], d["lam"], d["rows"], d["nnz"])      # bit for bit
    assert (q["rows"], q["nnz"], q["reason"]) == (d["rows"], d["nnz"], 2) and abs(q["its"] - d["its"]) <= 1
    assert np.allclose(q["lam"], d["lam"], rtol=1e-12) and abs(q["xs"] - d["xs"]) <= 1e-8 * d["xs"]


def test_gamg_reasons_and_reuse():
    """KSP reasons through the gamg loop (0 iterations on b = 0, the iteration limit), a second solve with NEW values on
    the same pattern (aggregates reused, Galerkin sums redone: the solution scales with the operator), and a new pattern
    (hierarchy rebuilt)."""
    mesh = H.gen_box_tets(-1, 1, 14, -1, 1, 12, -1, 1, 10)
    s, dm = _device_problem(pf.POISSON_TET, mesh, H.POISSON_ELEMDATA)
    s.setPreconditioner("gamg")
    s.setTolerances(rtol=1e-10)
    its, reason, _ = s.factoriseAndSolve()
    x1 = s.getSolution()
    sym1 = s.amgInfo()["symbolic_ms"]
    assert reason == 2 and its < 25
    s.setTolerances(rtol=1e-10, maxits=2)
    assert s.factoriseAndSolve()[:2] == (2, -3)
    s.setTolerances(rtol=1e-10)
    s.assemble(np.array([2.0, 2.0, 2.0]), H.TIMEDATA)              # kx = ky = kz = 2: K doubles; so does the lifted part of F
    rhs2 = s.getRHS()
    its2, reason2, _ = s.factoriseAndSolve()
    assert reason2 == 2 and abs(its2 - its) <= 1 and s.amgInfo()["symbolic_ms"] == sym1          # same aggregates
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    rowptr, cols, vals = s.getCSR()
    u2 = spl.spsolve(sp.csr_matrix((vals, cols, rowptr)).tocsc(), rhs2)
    assert np.abs(s.getSolution() - u2).max() <= 1e-8 * np.abs(u2).max() and np.abs(x1).max() > 0
    s.buildPattern()                                                # a new pattern: the hierarchy is rebuilt
    s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
    its3, reason3, _ = s.factoriseAndSolve()
    assert (its3, reason3) == (its, 2) and np.array_equal(s.getSolution(), x1)
    # b = 0 (the beam without its body force): converged before the first iteration, KSP_CONVERGED_ATOL
    ed0 = H.ELAST_ELEMDATA.copy()
    ed0[3:] = 0.0
    z, _ = _device_problem(pf.ELAST_TET, H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3), ed0)
    z.setPreconditioner("gamg")
    assert z.factoriseAndSolve()[:2] == (0, 3) and not z.getSolution().any()


@pytest.mark.parametrize("case", ["cube40", "beam", "cube40_w", "beam_w"])
def test_gamg_fused_cycle_equals_level_by_level_kernels(case, beam, monkeypatch):
    """The V-cycle as it is run -- SpMV + vector step in one kernel on the coarse levels (k_amg_spmv_ep), step 0 of a level's
    pre-smoothing inside the restriction above it, every level of at most 4096 rows in ONE launch (k_amg_tail) -- against
    the same cycle enqueued level by level (PFEM_AMG_FUSED=0: one kernel per operation): the same arithmetic in the same
    order, so the residual history and the solution are equal bit for bit; and as a hipGraph replay or plain launches."""
    w = case.endswith("_w")          # the W-cycle's second visits too (the step 0 a fused level expects comes from k_amg_w_between there)
    case = case[:-2] if w else case
    kind, mesh, ed = ((pf.POISSON_TET, H.gen_box_tets(-1, 1, 40, -1, 1, 40, -1, 1, 40), H.POISSON_ELEMDATA) if case == "cube40" else
                      (pf.ELAST_TET, H.gen_box_tets(-0.5, 0.5, 8, 0.0, 6.0, 48, -0.5, 0.5, 8, bc_mode=1, ndof=3), H.ELAST_ELEMDATA))
    out = {}
    for fused, graph in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("PFEM_AMG_FUSED", fused)
        monkeypatch.setenv("PFEM_CG_GRAPH", graph)
        s, _ = _device_problem(kind, mesh, ed)
        if case == "beam":
            s.setSpmvFormat("grouped")      # node-wise aggregation (the 3 dof rows of a node as one group) at this size too
            s.buildPattern()
            s.assemble(ed, H.TIMEDATA)
        s.setPreconditioner("gamg")
        if w:
            s.setAmgCycle("w")
        s.setTolerances(rtol=1e-10, maxits=5000)
        its, reason, _ = s.factoriseAndSolve()
        assert reason == 2 and s.amgCycle()["cycle"] == ("w" if w else "v")
        out[fused, graph] = (its, s.getHistory(), s.getSolution(), s.amgInfo()["rows"])
        s.free()
    a = out["1", "1"]
    assert len(a[3]) >= 3
    for key in (("0", "1"), ("1", "0")):
        b = out[key]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_gamg_bound_and_diagonal_from_the_assembly_kernel_keep_every_bit(monkeypatch):
    """From the second step of a pattern on, the Poisson gather kernel leaves level 0's inverse diagonal and the rows' Gershgorin
    ratios behind while the rows are still in LDS, and the multigrid's numeric phase no longer reads the assembled matrix for
    them (k_amg_diag_bound on level 0: 1.4 GB at config 3).  Same diagonal, same bound (a maximum: order-free), hence the same
    residual history and iterate, bit for bit -- against the first step (which has no hierarchy yet and takes the old path) and
    against PFEM_DEBUG_NO_ASM_BOUND=1; new values on the same pattern follow."""
    mesh = H.gen_box_tets(-1, 1, 36, -1, 1, 30, -1, 1, 33)
    ed, ed2 = H.POISSON_ELEMDATA, np.array([1.3, 0.7, 2.1])
    out = {}
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("PFEM_DEBUG_NO_ASM_BOUND", off)
        else:
            monkeypatch.delenv("PFEM_DEBUG_NO_ASM_BOUND", raising=False)
        s, _ = _device_problem(pf.POISSON_TET, mesh, ed)
        s.setPreconditioner("gamg")
        s.setTolerances(rtol=1e-10, maxits=500)
        runs = []
        for data in (ed, ed, ed2, ed):
            s.assemble(data, H.TIMEDATA)
            its, reason, _ = s.factoriseAndSolve()
            assert reason == 2
            runs.append((its, s.getHistory(), s.getSolution(), list(s.amgInfo()["lambda_max"])))
        out[off] = runs
        s.free()
    a, b = out[""], out["1"]
    for ra, rb in zip(a, b):
        assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2]) and ra[3] == rb[3]
    # the first step (old path) and the second and fourth (new path) saw the same values; the third saw others
    assert np.array_equal(a[0][1], a[1][1]) and np.array_equal(a[0][2], a[3][2]) and a[0][3] == a[1][3]
    assert not np.array_equal(a[0][2], a[2][2])


@pytest.mark.parametrize("kind_name", ["poisson", "elast"])
def test_incidence_lists_as_translated_patterns_keep_every_bit(kind_name, monkeypatch):
    """A box numbered along its lines: the gather kernels read a node's incidence list from the pattern it is a translated copy
    of (k_incpat_*: interior, faces, edges, corners of the constrained boundary -- a few dozen lists) instead of 16 B per visit
    of its own (two thirds of the Poisson kernel's HBM fetches).  K and F are the same bits as with every node's own records
    (PFEM_DEBUG_INC_OWN_RECORDS=1, looked up at every assembly) and as the oracle's serial loop; a mesh under a random numbering
    has as many patterns as nodes, keeps its own records and assembles the same matrix."""
    if kind_name == "poisson":
        kind, ed = pf.POISSON_TET, np.array([1.3, 0.7, 2.1])
        mesh = H.gen_box_tets(-1, 1, 21, -1, 1, 17, -1, 1, 19)
    else:
        kind, ed = pf.ELAST_TET, H.ELAST_ELEMDATA
        mesh = H.gen_box_tets(-0.5, 0.5, 7, 0.0, 6.0, 30, -0.5, 0.5, 6, bc_mode=1, ndof=3)
    monkeypatch.delenv("PFEM_DEBUG_INC_OWN_RECORDS", raising=False)
    s, dm = _device_problem(kind, mesh, ed)
    n_pat, longest = s.incidencePatterns()
    assert 1 <= n_pat <= 200 and 1 <= longest <= 24, (n_pat, longest)
    rowptr, cols, vals = s.getCSR()
    rhs = s.getRHS()
    monkeypatch.setenv("PFEM_DEBUG_INC_OWN_RECORDS", "1")
    s.assemble(ed, H.TIMEDATA)
    _, _, vals_own = s.getCSR()
    assert np.array_equal(vals, vals_own) and np.array_equal(rhs, s.getRHS())
    monkeypatch.delenv("PFEM_DEBUG_INC_OWN_RECORDS")
    s.assemble(ed, H.TIMEDATA)
    assert np.array_equal(vals, s.getCSR()[2]) and np.array_equal(rhs, s.getRHS())
    prob = O.setup_problem(kind, _omesh(mesh), elemData=ed)
    assert np.array_equal(prob.rowptr, rowptr) and np.array_equal(prob.cols, cols)
    assert np.array_equal(prob.vals, vals) and np.array_equal(prob.rhs, rhs)
    s.free()
    s2, _ = _device_problem(kind, _shuffled(mesh), ed)
    assert s2.incidencePatterns() == (0, 0)
    s2.free()


def _shuffled(mesh, seed=7):
    """The same mesh under a random node numbering (what an arbitrary mesh file may look like)."""
    perm = np.random.default_rng(seed).permutation(mesh.nNode).astype(np.int32)       # old id -> new id
    xyz = np.empty_like(mesh.xyz)
    xyz[:, perm] = mesh.xyz
    return H.Mesh(xyz, perm[mesh.conn], perm[mesh.bc_node], mesh.bc_dof, mesh.bc_val, box=mesh.box)


@pytest.mark.parametrize("kind_name", ["poisson", "elast"])
def test_internal_renumbering_is_invisible_at_the_boundary(kind_name, monkeypatch):
    """A mesh whose node numbering has no locality is renumbered INSIDE the library (Morton order of the nodes, owned dofs
    only; taken automatically when an element's dofs lie far apart, PFEM_REORDER=0/1 forces it) -- and nothing of it shows
    at the ABI: pattern, K and F come back in the caller's numbering with the same bits as without the renumbering (and as
    the oracle's serial loop), SpMV / solution / element dof array / aggregates are indexed by the caller's dofs, the
    solve gives the same answer; and the SpMV form that the renumbered matrix allows is a 16-bit one again."""
    if kind_name == "poisson":
        kind, ed, okind = pf.POISSON_TET, H.POISSON_ELEMDATA, O.POISSON_TET
        mesh = _shuffled(H.gen_box_tets(-1, 1, 22, -1, 1, 20, -1, 1, 24))
    else:
        kind, ed, okind = pf.ELAST_TET, H.ELAST_ELEMDATA, O.ELAST_TET
        mesh = _shuffled(H.gen_box_tets(-0.5, 0.5, 8, 0.0, 6.0, 40, -0.5, 0.5, 8, bc_mode=1, ndof=3))
    out = {}
    for reorder in ("0", "auto"):
        if reorder == "auto":
            monkeypatch.delenv("PFEM_REORDER", raising=False)
        else:
            monkeypatch.setenv("PFEM_REORDER", reorder)
        s, dm = _device_problem(kind, mesh, ed)
        rowptr, cols, vals = s.getCSR()
        rng = np.random.default_rng(3)
        x = rng.standard_normal(dm.size_global)
        conn_d, xyz_d, edof_l, soln_d = s.downloadMesh()
        s.setTolerances(rtol=1e-10, maxits=20000)
        its, reason, _ = s.factoriseAndSolve()
        u = s.getSolution()
        s.setPreconditioner("gamg")
        its_g, reason_g, _ = s.factoriseAndSolve()
        ug = s.getSolution()
        agg0 = s.amgAggregates(0, dm.size_global)
        tr0 = s.amgTransfer(0)
        out[reorder] = dict(tr0=tr0, conn=conn_d, xyz=xyz_d, soln=soln_d, rowptr=rowptr, cols=cols, vals=vals, rhs=s.getRHS(), y=s.spmv(x), edof=edof_l, its=its, u=u, its_g=its_g, ug=ug,
                            agg0=agg0, bits=s.spmvColumnBits(), reasons=(reason, reason_g))
        s.free()
    a, b = out["0"], out["auto"]
    for k in ("rowptr", "cols", "vals", "rhs", "edof", "conn", "xyz", "soln"):        # (the nodes are renumbered inside too: invisible as well)
        assert np.array_equal(a[k], b[k]), k
    prob = O.setup_problem(okind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val))
    assert np.array_equal(b["rowptr"], prob.rowptr) and np.array_equal(b["cols"], prob.cols)
    assert np.array_equal(b["vals"], prob.vals) and np.array_equal(b["rhs"], prob.rhs)
    scale = np.abs(a["y"]).max()
    assert np.abs(a["y"] - b["y"]).max() <= 1e-12 * scale                      # a row sums its products in another order
    assert a["reasons"] == b["reasons"] == (2, 2) and abs(a["its"] - b["its"]) <= max(2, a["its"] // 50)
    assert np.abs(a["u"] - b["u"]).max() <= 1e-8 * max(1.0, np.abs(a["u"]).max())
    assert np.abs(b["ug"] - b["u"]).max() <= 1e-8 * max(1.0, np.abs(a["u"]).max())
    # the aggregates are indexed by the caller's dofs; the multigrid solve is at least as good as on the scrambled numbering
    assert b["bits"] == 16 and b["its_g"] <= a["its_g"] + 2
    if b["tr0"]["rbm"]:      # rigid-body modes: the map names the translation dofs of the aggregates (3 of every coarse node's 6)
        cb = b["tr0"]["coarse_bs"]
        assert cb == 6 and set(np.unique(b["agg0"] % cb)) == {0, 1, 2} and len(np.unique(b["agg0"] // cb)) == b["agg0"].max() // cb + 1
    else:
        assert len(np.unique(b["agg0"])) == b["agg0"].max() + 1
