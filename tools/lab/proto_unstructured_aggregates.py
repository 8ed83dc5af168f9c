#!/usr/bin/env python3
"""CPU experiment (oracle only, no device): which node aggregates make the rigid-body hierarchy converge on a beam whose
nodes have been moved off their lattice?  Compares, on the same jittered matrix: (a) the lattice's bricks, (b) pairwise
handshake matching on the strength graph (what the product falls back to without a lattice), (c) boxes of a uniform grid
over the coordinates (cell = 2 mean spacings, doubled per level).  Prints CG iterations at rtol 1e-5."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import pfem_oracle as O
import scipy.sparse as sp

nx, ny, nz = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (10, 60, 10)))
jitter = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
ext = (-.5, .5, 0.0, ny / nx, -.5, .5)          # cubes of edge 1/nx
mesh = O.gen_box_tets(ext[0], ext[1], nx, ext[2], ext[3], ny, ext[4], ext[5], nz, bc_mode=1, ndof=3)
h = 1.0 / nx
if jitter > 0:
    rng = np.random.default_rng(7)
    d = rng.uniform(-1, 1, size=mesh.xyz.shape) * (jitter * h / np.sqrt(3.0))
    d[:, np.unique(mesh.bc_node)] = 0.0
    mesh_j = O.Mesh(mesh.xyz + d, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val)
else:
    mesh_j = mesh
P0 = O.setup_problem(O.ELAST_TET, mesh)
PJ = O.setup_problem(O.ELAST_TET, mesh_j)
N = PJ.dm.size_global
assy = O.assy_for_soln(PJ.dm.NodeDofArrayNew)
node_of_dof = assy // 3
assert (node_of_dof[0::3] == node_of_dof[2::3]).all()
nodes = node_of_dof[0::3]
xyz_l = P0.xyz_new[:, nodes]        # lattice coordinates of the free nodes
xyz_j = PJ.xyz_new[:, nodes]
nn = xyz_l.shape[1]
print(f"beam {nx}x{ny}x{nz}: {nn} free nodes, {N} dofs, jitter {jitter}")
A = sp.csr_matrix((PJ.vals, PJ.cols, PJ.rowptr), shape=(N, N))


def node_graph(A, bs):
    """weights between nodes: Frobenius norm of the block"""
    C = A.tocoo()
    W = sp.csr_matrix((C.data ** 2, (C.row // bs, C.col // bs)), shape=(A.shape[0] // bs, A.shape[0] // bs))
    W.sum_duplicates()
    W.data = np.sqrt(W.data)
    return W


def renumber(a):
    _, inv = np.unique(a, return_inverse=True)
    return inv


def agg_bricks(pos, level):
    """bricks of 2^(level+1) positions per axis; a last thin brick joins its neighbour"""
    key = []
    for d in range(3):
        p = pos[d]
        s = 1 << (level + 1)
        b = p // s
        hi = p.max()
        if hi // s > 0 and (hi % s) < s // 2:
            b = np.minimum(b, hi // s - 1)
        key.append(b)
    return renumber(key[0] + 4096 * (key[1] + 4096 * key[2]))


def agg_match(W, passes=3, rounds=8):
    """handshake matching: every free node proposes to its strongest free neighbour, mutual proposals pair; leftovers join the
    aggregate of their strongest neighbour"""
    n = W.shape[0]
    agg = np.arange(n)
    for _ in range(passes):
        Wc = W.copy(); Wc.setdiag(0); Wc.eliminate_zeros(); Wc = Wc.tocsr()
        m = Wc.shape[0]
        mate = -np.ones(m, dtype=np.int64)
        for _r in range(rounds):
            free = mate < 0
            C = Wc.tocoo()
            ok = free[C.row] & free[C.col]
            if not ok.any():
                break
            r, c, v = C.row[ok], C.col[ok], C.data[ok]
            order = np.lexsort((c, -v, r))
            r, c = r[order], c[order]
            first = np.r_[True, r[1:] != r[:-1]]
            prop = -np.ones(m, dtype=np.int64)
            prop[r[first]] = c[first]
            has = prop >= 0
            idx = np.where(has)[0]
            mutual = idx[prop[prop[idx]] == idx]
            mate[mutual] = prop[mutual]
        # leftovers join strongest neighbour's pair
        new = -np.ones(m, dtype=np.int64)
        paired = np.where(mate >= 0)[0]
        lead = paired[paired < mate[paired]]
        new[lead] = np.arange(len(lead)); new[mate[lead]] = new[lead]
        k = len(lead)
        left = np.where(mate < 0)[0]
        C = Wc.tocoo()
        for i in left:
            lo, hi_ = Wc.indptr[i], Wc.indptr[i + 1]
            nb = Wc.indices[lo:hi_]; wv = Wc.data[lo:hi_]
            cand = [(w, j) for w, j in zip(wv, nb) if mate[j] >= 0]
            if cand:
                new[i] = new[max(cand)[1]]
            else:
                new[i] = k; k += 1
        agg = new[agg]
        Pm = sp.csr_matrix((np.ones(m), (np.arange(m), new)), shape=(m, k))
        W = (Pm.T @ W @ Pm).tocsr()
    return renumber(agg)


def agg_grid(xyz, cell, origin):
    key = [np.floor((xyz[d] - origin[d]) / cell).astype(np.int64) for d in range(3)]
    return renumber(key[0] + 4096 * (key[1] + 4096 * key[2]))


def hierarchy(kind):
    Ps = []
    Al = A
    bs = 3
    cen = xyz_j
    pos = np.round((xyz_l - xyz_l.min(axis=1, keepdims=True)) / h).astype(np.int64)
    lvl = 0
    rows = [N]
    origin = xyz_j.min(axis=1) - 1e-9
    while Al.shape[0] > 128 and lvl < 10:
        nnodes = Al.shape[0] // bs
        if kind == "bricks":
            na = agg_bricks(pos, 0)
        elif kind == "match":
            na = agg_match(node_graph(Al, bs))
        elif kind == "grid":
            na = agg_grid(cen, 2.0 * h * (1 << lvl), origin)
        elif kind == "grid-half":
            na = agg_grid(cen, 2.0 * h * (1 << lvl), origin - 0.5 * h)
        elif kind == "grid-rand":
            na = agg_grid(cen, 2.0 * h * (1 << lvl), origin - np.array([0.37, 0.81, 0.13]) * h)
        if na.max() + 1 >= nnodes:
            break
        P, cen_new = O.rbm_prolongator(na, cen, 3, bs)
        Ps.append(P)
        Al = (P.T @ Al @ P).tocsr()
        dg = Al.diagonal()
        if (dg == 0).any():
            Al = (Al + sp.diags((dg == 0).astype(float))).tocsr()
        if kind == "bricks":
            # coarse node position = min position of members // 2
            npos = np.zeros((3, na.max() + 1), dtype=np.int64)
            for d in range(3):
                t = np.full(na.max() + 1, 1 << 30); np.minimum.at(t, na, pos[d]); npos[d] = t // 2
            pos = npos
        cen = cen_new
        bs = 6
        lvl += 1
        rows.append(Al.shape[0])
    return Ps, rows


for kind in (sys.argv[5].split(",") if len(sys.argv) > 5 else ("bricks", "match", "grid", "grid-half", "grid-rand")):
    Ps, rows = hierarchy(kind)
    for gamma, gfrom, gto, scale in ((1, 1, 99, 1.8), (2, 1, 99, 1.8), (2, 1, 99, 1.5), (2, 1, 99, 1.3), (2, 1, 99, 1.0), (2, 1, 1, 1.3), (2, 1, 2, 1.3)):
        M = O.amg_cycle(PJ.rowptr, PJ.cols, PJ.vals, Ps, 2, 8.0, scale, 128, 8, 1, None, gamma, gfrom, gto)
        x, its, reason, rn, hist = O.pcg_with(PJ.rowptr, PJ.cols, PJ.vals, PJ.rhs, M, 1e-5)
        print(f"   {kind} gamma {gamma} levels {gfrom}..{gto} scale {scale}: {its} iterations, reason {reason}", flush=True)
    cnt = np.bincount(np.asarray(Ps[0].tocsr()[0::3, :].tocoo().col) // 6)
    print(f"{kind:7s}: {its:4d} iterations (reason {reason}), rows {rows}, level-0 aggregate sizes min/mean/max {cnt.min()}/{cnt.mean():.1f}/{cnt.max()}")
