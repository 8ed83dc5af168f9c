"""The oracle against the reference's own artefacts (CPU, no GPU):
golden Ke/Fe from the flang-compiled reference routines, the shipped mesh files, and the
known answers of SURVEY 8(c).  Bit-exact where the quantity is an integer or an element matrix."""
import gzip
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import pfem_oracle as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "elements.npz"))


@pytest.fixture(scope="module")
def tet10(golden_dir):
    return O.read_mesh(os.path.join(golden_dir, "input", "tet10"))


@pytest.fixture(scope="module")
def tria20(golden_dir):
    return O.read_mesh(os.path.join(golden_dir, "input", "tria20x20"))


def test_elements_bit_exact_vs_reference_golden(gold, tet10, tria20):
    K, F = O.eval_elems(O.POISSON_TET, tet10.xyz, tet10.conn, O.POISSON_ELEMDATA)
    assert np.array_equal(K, gold["tet10_poisson_K"]) and np.array_equal(F, gold["tet10_poisson_F"])
    K, F = O.eval_elems(O.ELAST_TET, tet10.xyz, gold["tet10_elast_conn"], O.ELAST_ELEMDATA)
    assert np.array_equal(K, gold["tet10_elast_K"]) and np.array_equal(F, gold["tet10_elast_F"])
    K, F = O.eval_elems(O.POISSON_TET, gold["rt_xyz"], gold["rt_conn"], gold["rt_aniso"])
    assert np.array_equal(K, gold["rt_poisson_K"]) and np.array_equal(F, gold["rt_poisson_F"])
    K, F = O.eval_elems(O.ELAST_TET, gold["rt_xyz"], gold["rt_conn"], gold["rt_elast_data"])
    assert np.array_equal(K, gold["rt_elast_K"]) and np.array_equal(F, gold["rt_elast_F"])
    K, F = O.eval_elems(O.POISSON_TRIA, tria20.xyz, tria20.conn, np.array([1.0, 1.0]))
    assert np.array_equal(K, gold["tria20_K"]) and np.array_equal(F, gold["tria20_F"])
    K, F = O.eval_elems(O.POISSON_TRIA, gold["rtri_xy"], gold["rtri_conn"], gold["rtri_data"])
    assert np.array_equal(K, gold["rtri_K"]) and np.array_equal(F, gold["rtri_F"])


def test_elast_tria_bit_exact_vs_reference_golden(gold, golden_dir):
    """2-D sibling: plane-stress P1 triangle on the shipped Cook's-membrane mesh."""
    cook = O.read_mesh(os.path.join(golden_dir, "input", "cookmembranetria32"))
    assert (cook.nNode, cook.nElem, len(cook.bc_node)) == (1089, 2048, 66)
    K, F = O.eval_elems(O.ELAST_TRIA, cook.xyz, cook.conn, gold["cook_elast_data"])
    assert np.array_equal(K, gold["cook_elast_K"]) and np.array_equal(F, gold["cook_elast_F"])
    assert np.abs(K - K.transpose(0, 2, 1)).max() < 1e-12 * np.abs(K).max()


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference + flang)")
def test_elements_bit_exact_vs_live_reference(tet10):
    for kind, ed in ((O.POISSON_TET, O.POISSON_ELEMDATA), (O.ELAST_TET, O.ELAST_ELEMDATA)):
        K, F = O.eval_elems(kind, tet10.xyz, tet10.conn[:, :500], ed)
        Kr, Fr = O.ref_eval_elems(kind, tet10.xyz, tet10.conn[:, :500], ed)
        assert np.array_equal(K, Kr) and np.array_equal(F, Fr)


def test_unit_tet_single_precision_literals():
    # SURVEY finding 4 / probe E.2: K(2,2) on genTetra's first tet = 1.66666671633720398E-01*3
    xyz = np.array([[0, 1, 1, 1.0], [0, 0, 1, 0.0], [0, 0, 0, 1.0]])
    K, F = O.eval_elems(O.POISSON_TET, xyz, np.arange(4, dtype=np.int32).reshape(4, 1), O.POISSON_ELEMDATA)
    assert K[0, 0, 0] == 0.16666667163372040 and K[0, 1, 1] == 0.5000000149011612
    assert F[0, 0] == -0.25000000745058060
    assert O.ELAST_ELEMDATA[0] == 240.56500244140625 and O.ELAST_ELEMDATA[1] == 0.30000001192092896


def test_negative_jacobian_is_an_error(tet10):
    conn = tet10.conn[:, :4].copy()
    conn[[0, 1], 2] = conn[[1, 0], 2]
    with pytest.raises(RuntimeError):
        O.eval_elems(O.POISSON_TET, tet10.xyz, conn, O.POISSON_ELEMDATA)


def test_generator_reproduces_shipped_tet10(tet10):
    g = O.gen_box_tets(-2, 2, 10, -1, 1, 10, -1, 1, 10)
    assert np.array_equal(g.xyz, tet10.xyz) and np.array_equal(g.conn, tet10.conn)
    assert np.array_equal(g.bc_node, tet10.bc_node) and np.array_equal(g.bc_val, tet10.bc_val)


def test_generator_reproduces_shipped_tet100_bcs(golden_dir):
    bc = np.loadtxt(gzip.open(os.path.join(golden_dir, "input", "tet100-DirichBC.dat.gz"), "rt"))
    g = O.gen_box_tets(-1, 1, 100, -1, 1, 100, -1, 1, 100)
    assert len(bc) == 60002 and g.nNode == 1030301 and g.nElem == 6000000
    assert np.array_equal(g.bc_node + 1, bc[:, 0].astype(np.int64))
    assert np.array_equal(g.bc_val, bc[:, 2])
    a, b, c, d = (g.xyz[:, g.conn[i, ::997]] for i in range(4))
    jac = np.einsum("ij,ij->j", a - c, np.cross((b - c).T, (d - c).T).T)
    assert (jac > 0).all()


def test_tet10_sizes_and_known_answer(tet10):
    p = O.setup_problem(O.POISSON_TET, tet10)
    N = p.dm.size_global
    assert (N, len(p.cols)) == (729, 9097)
    A = sp.csr_matrix((p.vals, p.cols, p.rowptr), shape=(N, N))
    assert abs(A - A.T).max() == 0.0
    u = spl.spsolve(A.tocsc(), p.rhs)
    exact = (tet10.xyz ** 2).sum(0)
    assert np.abs(O.full_solution(p, u)[:, 0] - exact).max() < 2e-7  # probe E.5: 1.2e-7
    x, its, reason, rn, hist = O.pcg_jacobi(p.rowptr, p.cols, p.vals, p.rhs, rtol=1e-10, hist_len=100)
    assert reason == 2 and np.abs(x - u).max() < 1e-8
    assert hist[0] > 0 and rn <= 1e-10 * hist[0]
    assert np.array_equal(O.spmv(p.rowptr, p.cols, p.vals, x), A @ x) or np.allclose(O.spmv(p.rowptr, p.cols, p.vals, x), A @ x, rtol=1e-14)


def test_tria20x20_known_answer(tria20):
    p = O.setup_problem(O.POISSON_TRIA_INLINE, tria20, elemData=np.array([1.0, 1.0, 0.0]))
    N = p.dm.size_global
    assert (N, len(p.cols)) == (361, 2377)
    A = sp.csr_matrix((p.vals, p.cols, p.rowptr), shape=(N, N))
    u = spl.spsolve(A.tocsc(), p.rhs)
    assert abs(u.sum() - 68.09843993245326) < 1e-10
    assert np.allclose(u[:3], [0.13364425, 0.26399773, 0.38785071], atol=5e-9)
    x, y = tria20.xyz
    exact = np.sin(np.pi * x) * (np.cosh(np.pi * y) - np.cosh(np.pi) / np.sinh(np.pi) * np.sinh(np.pi * y))
    assert abs(np.abs(O.full_solution(p, u)[:, 0] - exact).max() - 7.11e-4) < 1e-5   # probe E.6
    # the module routine and the inline element agree to rounding
    p2 = O.setup_problem(O.POISSON_TRIA, tria20, elemData=np.array([1.0, 1.0, 0.0]))
    assert np.abs(p2.vals - p.vals).max() < 1e-12


def test_partitioned_numbering_invariants(tet10):
    rng = np.random.default_rng(3)
    npid = rng.integers(0, 3, tet10.nNode).astype(np.int32)           # any partition, METIS-like input
    dm = O.dof_numbering(tet10.nNode, 1, tet10.bc_node, tet10.bc_dof, tet10.bc_val, 3, npid)
    assert np.array_equal(dm.node_map_get_new[dm.node_map_get_old], np.arange(tet10.nNode))
    for p in range(3):                                                # ranks concatenated, ascending old id
        seg = dm.node_map_get_old[dm.node_start[p]:dm.node_end[p]]
        assert (npid[seg] == p).all() and (np.diff(seg) > 0).all()
    free = dm.NodeDofArrayNew[dm.NodeDofArrayNew >= 0]
    assert np.array_equal(free, np.arange(dm.size_global)) and dm.size_global == 729
    assert dm.row_start[0] == 0 and dm.row_end[-1] == 729 and (dm.row_start[1:] == dm.row_end[:-1]).all()
    # the partitioned problem is a symmetric permutation of the serial one
    p1 = O.setup_problem(O.POISSON_TET, tet10)
    p3 = O.setup_problem(O.POISSON_TET, tet10, nParts=3, node_proc_id=npid)
    u1 = spl.spsolve(sp.csr_matrix((p1.vals, p1.cols, p1.rowptr)).tocsc(), p1.rhs)
    u3 = spl.spsolve(sp.csr_matrix((p3.vals, p3.cols, p3.rowptr)).tocsc(), p3.rhs)
    assert np.abs(O.full_solution(p1, u1) - O.full_solution(p3, u3)).max() < 1e-12


def test_beam_config4_small_cousin():
    m = O.gen_box_tets(-0.5, 0.5, 2, 0.0, 6.0, 12, -0.5, 0.5, 2, bc_mode=1, ndof=3)
    assert len(m.bc_node) == 3 * 9 and (m.bc_val == 0).all()
    p = O.setup_problem(O.ELAST_TET, m)
    N = p.dm.size_global
    A = sp.csr_matrix((p.vals, p.cols, p.rowptr), shape=(N, N))
    assert abs(A - A.T).max() < 1e-10 * abs(A).max()
    u = spl.spsolve(A.tocsc(), p.rhs)
    full = O.full_solution(p, u)
    tip = full[m.xyz[1] == 6.0]
    # cantilever under body force (0.1,0,0): beam theory qL^4/8EI = 0.81; the coarse P1
    # mesh locks, so only sign/order of magnitude is asserted here
    assert 0.05 < tip[:, 0].mean() < 0.9


def test_single_reduction_pcg_restatement(tet10, tria20):
    """orc_pcg_jacobi_single_reduction (PETSc's KSPCGUseSingleReduction recurrences) against the two-reduction loop and
    a direct solve: same solution, same stopping iteration to +-1, the same residual history to rounding, same reasons."""
    for p in (O.setup_problem(O.POISSON_TET, tet10), O.setup_problem(O.POISSON_TRIA_INLINE, tria20, elemData=np.array([1.0, 1.0, 0.0]))):
        N = p.dm.size_global
        u = spl.spsolve(sp.csr_matrix((p.vals, p.cols, p.rowptr), shape=(N, N)).tocsc(), p.rhs)
        for rtol in (1e-5, 1e-10):
            x2, its2, reason2, rn2, h2 = O.pcg_jacobi(p.rowptr, p.cols, p.vals, p.rhs, rtol=rtol, hist_len=400)
            x1, its1, reason1, rn1, h1 = O.pcg_jacobi_single_reduction(p.rowptr, p.cols, p.vals, p.rhs, rtol=rtol, hist_len=400)
            assert reason1 == reason2 == 2 and abs(its1 - its2) <= 1
            n = min(len(h1), len(h2))
            assert np.allclose(h1[:n], h2[:n], rtol=1e-6)
            if rtol == 1e-10:
                assert np.abs(x1 - u).max() < 1e-8
    p = O.setup_problem(O.POISSON_TET, tet10)
    assert O.pcg_jacobi_single_reduction(p.rowptr, p.cols, p.vals, p.rhs, rtol=1e-10, maxits=7)[1:3] == (7, -3)
    assert O.pcg_jacobi_single_reduction(p.rowptr, p.cols, p.vals, p.rhs, rtol=1e-10, maxits=0)[1:3] == (0, -3)
    assert O.pcg_jacobi_single_reduction(p.rowptr, p.cols, p.vals, 0 * p.rhs)[1:3] == (0, 3)
    assert O.pcg_jacobi_single_reduction(p.rowptr, p.cols, -p.vals, p.rhs)[2] == -8        # (r, M^-1 r) < 0
    v = p.vals.copy()
    v[p.rowptr[5]:p.rowptr[5 + 1]][p.cols[p.rowptr[5]:p.rowptr[6]] != 5] *= 40.0           # keeps the diagonal positive
    v2 = sp.csr_matrix((v, p.cols, p.rowptr), shape=(729, 729)); v2 = ((v2 + v2.T) * 0.5).tocsr(); v2.sort_indices()
    r1 = O.pcg_jacobi_single_reduction(v2.indptr.astype(np.int64), v2.indices.astype(np.int32), v2.data, p.rhs, rtol=1e-12)
    r2 = O.pcg_jacobi(v2.indptr.astype(np.int64), v2.indices.astype(np.int32), v2.data, p.rhs, rtol=1e-12)
    assert r1[2] == r2[2] == -10                                                              # indefinite matrix


def test_amg_pcg_restatement(tet10):
    """The oracle's restatement of the product's -pc_type gamg solve (oracle.pcg_amg: Galerkin operators from GIVEN
    aggregates, Chebyshev smoothing with the Gershgorin bound, V(1,1) cycle, dense solve at the bottom, PETSc's KSPCG
    stopping rule): an SPD preconditioner, so CG converges to the direct solution in fewer iterations than with the
    diagonal; with no coarse level at all the cycle is the Chebyshev polynomial of D^-1 A."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    prob = O.setup_problem(O.POISSON_TET, tet10)
    A = sp.csr_matrix((prob.vals, prob.cols, prob.rowptr))
    u = spl.spsolve(A.tocsc(), prob.rhs)
    _, its_j, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-10)
    n = A.shape[0]                       # 729 = 9^3 free nodes: bricks of 2x2x2 nodes, then pairs of bricks
    i = np.arange(n)
    brick = ((i // 81) // 2) * 25 + (((i // 9) % 9) // 2) * 5 + (i % 9) // 2
    aggs = [brick, np.arange(125) // 2]
    x, its, reason, rn, hist = O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, aggs, rtol=1e-10)
    assert reason == 2 and its < its_j // 2 and len(hist) == its + 1 and hist[-1] == rn <= 1e-10 * hist[0]
    assert np.abs(x - u).max() <= 1e-9 * np.abs(u).max()
    x1, its1, reason1, *_ = O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, [], rtol=1e-10)
    assert reason1 == 2 and its < its1 < its_j and np.abs(x1 - u).max() <= 1e-9 * np.abs(u).max()
    # stopping rule: zero right-hand side, iteration limit
    assert O.pcg_amg(prob.rowptr, prob.cols, prob.vals, 0 * prob.rhs, aggs)[1:3] == (0, 3)
    assert O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, aggs, rtol=1e-14, maxits=3)[1:3] == (3, -3)


def test_lattice_brick_aggregates_restatement(tet10):
    """O.lattice_brick_aggregates (what the device's `bricks in one step` is compared with): from the node coordinates alone,
    a hierarchy of bricks -- 2x2x2 nodes per aggregate, the node an odd line leaves over in a thin brick of its own, every
    level aligned with the lattice of ALL mesh nodes (the Dirichlet planes count: the first free node of a line sits at
    position 1 and stays alone) --; the cycle built on it converges in a fraction of point Jacobi's iterations; a plane problem
    halves x, y, x, then y, x, y."""
    prob = O.setup_problem(O.POISSON_TET, tet10)
    free = np.where(prob.dm.NodeDofArrayNew.reshape(-1) >= 0)[0]
    aggs = O.lattice_brick_aggregates(prob.xyz_new, prob.xyz_new[:, free])
    i = np.arange(729)               # free node i sits at position (1 + i % 9, 1 + (i // 9) % 9, 1 + i // 81): position 0 is the Dirichlet plane
    brick = ((i // 81 + 1) // 2) * 25 + (((i // 9) % 9 + 1) // 2) * 5 + (i % 9 + 1) // 2
    assert len(aggs) == 1 and np.array_equal(aggs[0], brick)          # 729 -> 125 (<= 128: the dense bottom)
    mesh = O.gen_box_tets(-1, 1, 30, -1, 1, 30, -1, 1, 30)
    p30 = O.setup_problem(O.POISSON_TET, mesh)
    free = np.where(p30.dm.NodeDofArrayNew.reshape(-1) >= 0)[0]
    aggs = O.lattice_brick_aggregates(p30.xyz_new, p30.xyz_new[:, free])
    assert [len(a) for a in aggs] == [29 ** 3, 15 ** 3, 8 ** 3] and [int(a.max()) + 1 for a in aggs] == [15 ** 3, 8 ** 3, 4 ** 3]
    cnt = np.bincount(aggs[0])
    assert cnt.max() == 8 and cnt.min() == 1 and (cnt == 8).sum() == 14 ** 3          # 29 = 14 pairs + one left over, per axis
    x, its, reason, *_ = O.pcg_amg(p30.rowptr, p30.cols, p30.vals, p30.rhs, aggs, eig_ratio=16.0, rtol=1e-10)
    _, its_j, *_ = O.pcg_jacobi(p30.rowptr, p30.cols, p30.vals, p30.rhs, rtol=1e-10)
    assert reason == 2 and its < its_j // 4
    # -pc_mg_cycle_type w (gamma = 2: every coarse problem that is not the last visited twice, the second time on the residual of
    # the first): still a symmetric positive definite preconditioner (CG converges, reason 2), never more iterations than the
    # V-cycle, the same answer; restricted to level 1 (gamma_to) it lies between the two
    xw, its_w, reason_w, *_ = O.pcg_amg(p30.rowptr, p30.cols, p30.vals, p30.rhs, aggs, eig_ratio=16.0, rtol=1e-10, gamma=2)
    xw1, its_w1, reason_w1, *_ = O.pcg_amg(p30.rowptr, p30.cols, p30.vals, p30.rhs, aggs, eig_ratio=16.0, rtol=1e-10, gamma=2, gamma_to=1)
    assert reason_w == 2 and reason_w1 == 2 and its_w <= its_w1 <= its and its_w < its
    assert np.abs(xw - x).max() <= 1e-8 * np.abs(x).max() and np.abs(xw1 - x).max() <= 1e-8 * np.abs(x).max()
    # the plane: 19 x 19 free nodes of the tria20x20 mesh: x, y, x halved on the first level (aggregates of 4 x 2 nodes)
    g = np.meshgrid(np.arange(21.0), np.arange(21.0), indexing="ij")
    xy = np.stack([g[0].ravel(), g[1].ravel()])
    inner = (xy[0] > 0) & (xy[0] < 20) & (xy[1] > 0) & (xy[1] < 20)
    a2 = O.lattice_brick_aggregates(xy, xy[:, inner])
    assert len(a2[0]) == 361 and np.bincount(a2[0]).max() == 8 and int(a2[0].max()) + 1 == 5 * 10


def test_rigid_body_prolongator_restatement():
    """O.rbm_prolongator: the columns of an aggregate span its rigid-body motions -- a motion T + W x (x - c) of the whole mesh
    is reproduced exactly from the coarse vector (T + W x c_I, W) of every aggregate, on the level of the assembled matrix
    (3 dofs per node) and on a level of 6-dof nodes; an aggregate of collinear nodes keeps its translations only.  With these
    prolongators the cycle needs a small fraction of the iterations the translations alone need on a slender beam."""
    rng = np.random.default_rng(5)
    n = 60
    xyz = rng.standard_normal((3, n))
    agg = np.repeat(np.arange(n // 6), 6)
    P, cen = O.rbm_prolongator(agg, xyz, 3, 3)
    assert P.shape == (3 * n, 6 * (n // 6))
    T, W = np.array([0.3, -1.2, 0.7]), np.array([0.5, 0.25, -2.0])
    coarse = np.concatenate([np.concatenate([T + np.cross(W, cen[:, a]), W]) for a in range(n // 6)])
    u = (T[:, None] + np.cross(W, xyz.T).T).T.ravel()
    assert np.abs(P @ coarse - u).max() <= 1e-13
    # the next level: nodes of 6 dofs (the centroids), aggregates of 5
    agg2 = np.repeat(np.arange(2), 5)
    P2, cen2 = O.rbm_prolongator(agg2, cen, 3, 6)
    coarse2 = np.concatenate([np.concatenate([T + np.cross(W, cen2[:, a]), W]) for a in range(2)])
    assert np.abs(P2 @ coarse2 - coarse).max() <= 1e-13
    # three collinear nodes: no rotations (offsets 0), translations intact
    line = np.stack([np.arange(3.0), 2 * np.arange(3.0), np.zeros(3)])
    Pl, _ = O.rbm_prolongator(np.zeros(3, int), line, 3, 3)
    assert abs(Pl[:, 3:]).sum() == 0 and np.array_equal(Pl[:, :3].toarray(), np.tile(np.eye(3), (3, 1)))
    # plane: 2 + 1 modes
    xy = np.vstack([rng.standard_normal((2, 8)), np.zeros((1, 8))])
    Pp, cp = O.rbm_prolongator(np.repeat([0, 1], 4), xy, 2, 2)
    wz = 0.8
    up = np.stack([T[0] - wz * xy[1], T[1] + wz * xy[0]]).T.ravel()
    cz = np.concatenate([[T[0] - wz * cp[1, a], T[1] + wz * cp[0, a], wz] for a in range(2)])
    assert Pp.shape == (16, 6) and np.abs(Pp @ cz - up).max() <= 1e-13
    # a slender clamped beam: bricks of nodes, with and without the rotations
    mesh = O.gen_box_tets(-0.5, 0.5, 4, 0.0, 6.0, 24, -0.5, 0.5, 4, bc_mode=1, ndof=3)
    prob = O.setup_problem(O.ELAST_TET, mesh)
    nd = prob.dm.NodeDofArrayNew.reshape(-1, 3)
    free = np.where(nd[:, 0] >= 0)[0]
    x0 = prob.xyz_new[:, free]
    node_aggs = O.lattice_brick_aggregates(prob.xyz_new, x0, dense_limit=20)
    Ps, xs, fb = [], x0, 3
    for a in node_aggs:
        Pk, xs = O.rbm_prolongator(a, xs, 3, fb)
        Ps.append(Pk)
        fb = 6
    dof_aggs = [np.repeat(3 * a, 3) + np.tile(np.arange(3), len(a)) for a in node_aggs]
    _, its_r, reason_r, *_ = O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, Ps, eig_ratio=16.0)
    _, its_t, reason_t, *_ = O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, dof_aggs, eig_ratio=8.0, coarse_scale=1.8)
    assert (reason_r, reason_t) == (2, 2) and 3 * its_r < its_t, (its_r, its_t)


def test_node_brick_and_split_brick_restatements():
    """Round 6's two restatements.  O.lattice_node_brick_aggregates (what the device's node bricks in one step are compared with):
    boxes of nodes, 4 to the edge along an axis of 24 and more nodes on level 0 (3 from 6 on), pairs below, the short brick at the
    end of a line joined to its neighbour; with owners the bricks are cut where the owner changes and no aggregate holds nodes of
    two ranks; the rigid-body cycle built on them converges like the one on 2x2x2 bricks.  O.lattice_brick_aggregates(split=True)
    (amg_split_bricks): on a partition whose parts are not boxes an aggregate is one owner's part of a 2x2x2 brick -- the parts of
    a brick together ARE the one-rank brick."""
    mesh = O.gen_box_tets(-0.5, 0.5, 8, 0.0, 6.0, 48, -0.5, 0.5, 8, bc_mode=1, ndof=3)
    prob = O.setup_problem(O.ELAST_TET, mesh)
    nd = prob.dm.NodeDofArrayNew.reshape(-1, 3)
    free = np.where(nd[:, 0] >= 0)[0]
    x0 = prob.xyz_new[:, free]
    aggs = O.lattice_node_brick_aggregates(prob.xyz_new, x0)
    # 9 x 48 x 9 free nodes: x, z in bricks of 3 (9 >= 6), y in bricks of 4 (48 >= 24): 3 x 12 x 3 aggregates, then pairs
    # (level 1: 3 x 12 x 3 coarse nodes in pairs, the third of a line of three joined to the pair: 1 x 6 x 1)
    assert [int(a.max()) + 1 for a in aggs][:2] == [3 * 12 * 3, 1 * 6 * 1] and len(aggs[0]) == 9 * 48 * 9
    cnt = np.bincount(aggs[0])
    # (y positions 1 .. 48 in bricks aligned on multiples of 4: {1,2,3}, {4..7}, ..., {44..47} + the single 48 joined to it)
    assert sorted(set(cnt.tolist())) == [27, 36, 45]
    owner = np.minimum(np.arange(len(free)) * 3 // len(free), 2)
    # (nodes are numbered x fastest, then y, then z: thirds of the numbering are slabs across z -- boxes)
    own = O.lattice_node_brick_aggregates(prob.xyz_new, x0, owner=owner)
    assert own is not None and all(len(np.unique(owner[own[0] == a])) == 1 for a in range(int(own[0].max()) + 1))
    assert int(own[0].max()) + 1 >= int(aggs[0].max()) + 1
    Ps, xs, fb = [], x0, 3
    for a in aggs:
        Pk, xs = O.rbm_prolongator(a, xs, 3, fb)
        Ps.append(Pk)
        fb = 6
    _, its_r, reason_r, *_ = O.pcg_amg(prob.rowptr, prob.cols, prob.vals, prob.rhs, Ps, eig_ratio=16.0)
    _, its_j, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs)
    assert reason_r == 2 and its_r < 40 and 10 * its_r < its_j, (its_r, its_j)
    # an owner set that is no box (one node more to the last rank): no node bricks across the ranks
    bad = owner.copy()
    bad[5] = 2
    assert O.lattice_node_brick_aggregates(prob.xyz_new, x0, owner=np.sort(bad)) is None
    # ---- split bricks: a staircase partition of a cube
    cube = O.gen_box_tets(-1, 1, 12, -1, 1, 10, -1, 1, 14)
    z = cube.xyz[2]
    step = (cube.xyz[0] > 1e-9).astype(np.float64)
    layer = np.floor((z + 1.0) / 2.0 * 14 - 1e-9) - step
    npid = np.clip(np.floor(layer * 3 / 14), 0, 2).astype(np.int32)
    p3 = O.setup_problem(O.POISSON_TET, cube, nParts=3, node_proc_id=npid)
    nda = p3.dm.NodeDofArrayNew.reshape(-1)
    fr = np.where(nda >= 0)[0]
    own3 = np.zeros(len(fr), np.int64)
    for r in range(3):
        own3[int(p3.dm.row_start[r]):int(p3.dm.row_end[r])] = r
    assert O.lattice_brick_aggregates(p3.xyz_new, p3.xyz_new[:, fr], owner=own3) is None                # the parts are not boxes
    sp = O.lattice_brick_aggregates(p3.xyz_new, p3.xyz_new[:, fr], owner=own3, split=True, replicate_rows=0)
    one = O.lattice_brick_aggregates(p3.xyz_new, p3.xyz_new[:, fr])
    assert sp is not None and len(sp) >= 2
    # no aggregate holds dofs of two owners; the aggregates that share a one-rank brick make up exactly that brick
    a0 = sp[0]
    assert all(len(np.unique(own3[a0 == a])) == 1 for a in range(int(a0.max()) + 1))
    brick_of = np.full(int(a0.max()) + 1, -1)
    for i, a in enumerate(a0):
        assert brick_of[a] in (-1, one[0][i])
        brick_of[a] = one[0][i]
    assert int(one[0].max()) + 1 <= int(a0.max()) + 1 <= 1.1 * (int(one[0].max()) + 1)
    _, its_s, reason_s, *_ = O.pcg_amg(p3.rowptr, p3.cols, p3.vals, p3.rhs, sp, eig_ratio=16.0, rtol=1e-10)
    _, its_1, reason_1, *_ = O.pcg_amg(p3.rowptr, p3.cols, p3.vals, p3.rhs, one, eig_ratio=16.0, rtol=1e-10)
    assert (reason_s, reason_1) == (2, 2) and its_s <= its_1 + 2, (its_s, its_1)


def test_mpi_restatement_of_the_cpu_baseline_equals_the_serial_oracle():
    """oracle/pfem_oracle_mpi (bench.py's cpu_baseline with one MPI rank per core): slabs of node planes, the oracle's element
    routine, distributed Jacobi-PCG -- same matrix size, same iteration count, same residual norm and nodal error as the
    serial oracle on the whole mesh, on 1, 3 and 5 ranks."""
    import json
    import shutil
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "pfem_oracle_mpi")
    launcher = next((c for c in ("/opt/conda/bin/mpiexec", shutil.which("mpiexec")) if c and os.path.exists(c)), None)
    if not launcher or not os.path.exists(exe):
        pytest.skip("no MPI here (make -C oracle mpi builds the program where one is installed)")
    m = 12
    mesh = O.gen_box_tets(-1, 1, m, -1, 1, m, -1, 1, m)
    dm = O.dof_numbering(mesh.nNode, 1, mesh.bc_node, mesh.bc_dof, mesh.bc_val)
    edof = O.elem_dof_array(mesh.conn, dm.NodeDofArrayNew)
    rowptr, cols = O.csr_pattern(edof, dm.size_global)
    vals, rhs = O.assemble(O.POISSON_TET, mesh.xyz, mesh.conn, edof, dm.solnApplied, O.POISSON_ELEMDATA, dm.size_global, rowptr, cols)
    x, its, reason, rn, _ = O.pcg_jacobi(rowptr, cols, vals, rhs, rtol=1e-10)
    for ranks in (1, 3, 5):
        r = subprocess.run([launcher, "-n", str(ranks), exe, str(m), "1e-10", "10000", "1"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert (d["ranks"], d["free_dofs"], d["nnz"]) == (ranks, dm.size_global, len(cols))
        assert (d["iterations"], d["converged_reason"]) == (its, reason) and abs(d["rnorm"] - rn) <= 1e-9 * rn
        assert d["max_nodal_error"] < 2e-3           # P1 elements on a 12^3 mesh against u = x^2+y^2+z^2


def test_lattice_by_numbering_restatement():
    """The strides of a box's numbering from the element connectivity alone (the coordinates may be anything): the generated
    boxes number their nodes along x, then y, then z; under a random numbering no strides fit."""
    for nx, ny, nz in ((5, 4, 3), (3, 7, 4), (8, 8, 8)):
        mesh = O.gen_box_tets(0, 1, nx, 0, 1, ny, 0, 1, nz)
        got = O.lattice_by_numbering(mesh.conn, mesh.nNode)
        assert got is not None and (got[0], got[1]) == (nx + 1, (nx + 1) * (ny + 1))
        assert got[2].max(axis=1).tolist() == [nx, ny, nz]
        perm = np.random.default_rng(3).permutation(mesh.nNode)
        assert O.lattice_by_numbering(perm[mesh.conn], mesh.nNode) is None
