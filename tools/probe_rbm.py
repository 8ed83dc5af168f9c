"""CPU probe (scipy, no GPU): does a rigid-body-mode coarse space pay for the beam of config 4?

Geometric 2x2x2 brick aggregates on the node lattice (what the product's lattice pairing forms), V(1,1) Chebyshev cycle as
in oracle.amg_cycle, and three tentative prolongators:
  t   translations only (3 columns per aggregate; the round-3 product)
  r   translations + rotations about the aggregate's centroid (6 columns), unsmoothed
  s   the same, smoothed once with damped Jacobi (classic smoothed aggregation)
Prints iterations of PCG (rtol 1e-5, preconditioned norm) and the operator complexity.
usage: probe_rbm.py [scale=0.24] [kinds=t,r,s]"""
import os
import sys
import time
import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pfem_oracle as O   # noqa: E402


def beam(scale):
    nx, ny, nz = max(2, round(50 * scale)), max(4, round(300 * scale)), max(2, round(50 * scale))
    mesh = O.gen_box_tets(-0.5, 0.5, nx, 0.0, 6.0, ny, -0.5, 0.5, nz, bc_mode=1, ndof=3)
    prob = O.setup_problem(O.ELAST_TET, mesh)
    return mesh, prob, (nx, ny, nz)


def hierarchy(A, xyz_free, pos, kind, omega=0.0, dense_limit=128, bricks3=True):
    """levels: list of (A_l, P_l); nodes carry bs dofs: 3 on level 0, 3 ('t') or 6 ('r','s') below"""
    levels, Ps = [A], []
    bs = 3
    xyz = xyz_free
    while levels[-1].shape[0] > dense_limit and len(levels) < 12:
        n_nodes = len(pos)
        hi = pos.max(0)
        cpos = pos // 2
        if bricks3:          # the leftover node of an odd line joins the pair next to it (product: bricks of 3 on 3-dof problems)
            for d in range(3):
                if hi[d] % 2 == 0 and hi[d] >= 2:
                    cpos[:, d] = np.minimum(cpos[:, d], hi[d] // 2 - 1)
        key = (cpos[:, 2] * 4096 + cpos[:, 1]) * 4096 + cpos[:, 0]
        uk, agg = np.unique(key, return_inverse=True)
        na = len(uk)
        if na >= n_nodes:
            break
        cnt = np.bincount(agg, minlength=na).astype(float)
        cen = np.stack([np.bincount(agg, weights=xyz[:, d], minlength=na) / cnt for d in range(3)], 1)
        r = xyz - cen[agg]
        cbs = 3 if kind == "t" else 6
        rows, cols, vals = [], [], []
        node = np.arange(n_nodes)

        def put(c, a, v):
            rows.append(node * bs + c); cols.append(agg * cbs + a); vals.append(v * np.ones(n_nodes))
        for c in range(3):
            put(c, c, 1.0)
        if cbs == 6:
            # u = T + W x r :  u_x = Wy rz - Wz ry ; u_y = Wz rx - Wx rz ; u_z = Wx ry - Wy rx
            put(0, 4, r[:, 2]); put(0, 5, -r[:, 1])
            put(1, 5, r[:, 0]); put(1, 3, -r[:, 2])
            put(2, 3, r[:, 1]); put(2, 4, -r[:, 0])
            if bs == 6:
                for c in range(3):
                    put(3 + c, 3 + c, 1.0)
        P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_nodes * bs, na * cbs))
        Al = levels[-1]
        if kind == "s" and omega > 0:
            d = Al.diagonal()
            lam = float((abs(Al) @ np.ones(Al.shape[0]) / d).max())
            P = (P - (omega / lam) * sp.diags(1.0 / d) @ (Al @ P)).tocsr()
        Ac = (P.T @ Al @ P).tocsr()
        # rotation columns of a single-node (or collinear) aggregate are null: drop zero-diagonal coarse dofs
        dg = Ac.diagonal()
        keep = dg > 1e-14 * dg.max()
        if not keep.all():
            S = sp.identity(Ac.shape[0], format="csr")[:, keep]
            P = (P @ S).tocsr(); Ac = (S.T @ Ac @ S).tocsr()
            assert kind != "t"
            raise SystemExit("null rotation column: aggregate without extent; probe does not handle dropping with 6-blocks")
        Ps.append(P); levels.append(Ac)
        bs = cbs
        pos = np.stack([(uk % 4096), (uk // 4096) % 4096, uk // (4096 * 4096)], 1)
        xyz = cen
    return levels, Ps


def cycle_fn(levels, Ps, cheb_degree=2, fine_degree=1, eig_ratio=8.0, coarse_scale=1.8, dense_limit=128, coarsest_sweeps=8, lam_mode="gersh"):
    dinv, lam = [], []
    for Al in levels:
        d = Al.diagonal()
        dinv.append(1.0 / d)
        if lam_mode == "gersh":
            lam.append(float((abs(Al) @ np.ones(Al.shape[0]) / d).max()))
        elif lam_mode == "gsym":          # Gershgorin on D^-1/2 A D^-1/2: invariant under a scaling of the basis
            sq = 1.0 / np.sqrt(d)
            lam.append(float(((abs(Al) @ sq) * sq).max()))
        else:
            v = np.random.default_rng(0).standard_normal(Al.shape[0])
            for _ in range(30):
                v = (Al @ v) / d; v /= np.linalg.norm(v)
            lam.append(1.1 * float(v @ ((Al @ v) / d)))
    dense = levels[-1].shape[0] <= max(dense_limit, 768)
    Ainv = np.linalg.inv(levels[-1].toarray()) if dense else None

    def smooth(l, x, rhs, deg):
        Al, d = levels[l], dinv[l]
        lmax = lam[l]; lmin = lmax / eig_ratio
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta; rho = 1.0 / sigma
        r = rhs.copy() if x is None else rhs - Al @ x
        dd = d * r / theta
        x = dd.copy() if x is None else x + dd
        for _ in range(1, deg):
            r = r - Al @ dd
            rho_new = 1.0 / (2.0 * sigma - rho)
            dd = rho_new * rho * dd + (2.0 * rho_new / delta) * (d * r)
            x = x + dd; rho = rho_new
        return x

    def cycle(l, rhs):
        if l == len(levels) - 1:
            return Ainv @ rhs if dense else smooth(l, None, rhs, coarsest_sweeps)
        deg = fine_degree if (l == 0 and fine_degree) else cheb_degree
        x = smooth(l, None, rhs, deg)
        rc = Ps[l].T @ (rhs - levels[l] @ x)
        x = x + coarse_scale * (Ps[l] @ cycle(l + 1, rc))
        return smooth(l, x, rhs, deg)
    return (lambda r: cycle(0, r)), lam


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.24
    kinds = (sys.argv[2] if len(sys.argv) > 2 else "t,r,s").split(",")
    t0 = time.time()
    mesh, prob, nE = beam(scale)
    N = prob.dm.size_global
    A = sp.csr_matrix((prob.vals, prob.cols, prob.rowptr), shape=(N, N))
    # free nodes in the solver's numbering: node of dof 3i is i-th free node; lattice positions from coordinates
    nd = prob.dm.NodeDofArrayNew.reshape(-1, 3) if hasattr(prob.dm, "NodeDofArrayNew") else None
    free_nodes = np.where(nd[:, 0] >= 0)[0]
    assert (nd[free_nodes, 0] == 3 * np.arange(len(free_nodes))).all()
    xyz_new = prob.xyz_new if hasattr(prob, "xyz_new") else mesh.xyz
    xyz = np.asarray(xyz_new).reshape(3, -1).T[free_nodes]
    pos = np.stack([np.unique(xyz[:, d], return_inverse=True)[1] for d in range(3)], 1)
    print(f"beam {nE} N={N} nnz={A.nnz} setup {time.time()-t0:.1f}s")
    for kind in kinds:
        for omega, scale_c, ratio, lam_mode in {
            "t": [(0, 1.8, 8.0, "gersh")],
            "r": [(0, 1.8, 8.0, "gersh"), (0, 1.0, 8.0, "gersh"), (0, 1.8, 8.0, "power"), (0, 1.5, 8.0, "power"), (0, 1.0, 8.0, "power"), (0, 1.8, 8.0, "gsym"), (0, 1.5, 8.0, "gsym"), (0, 1.0, 8.0, "gsym"), (0, 1.5, 16.0, "gsym")],
            "s": [(4.0 / 3.0, 1.0, 8.0, "gersh"), (4.0 / 3.0, 1.0, 8.0, "power"), (2.0 / 3.0, 1.0, 8.0, "gersh")],
        }[kind]:
            t1 = time.time()
            levels, Ps = hierarchy(A, xyz, pos, kind, omega)
            M, lam = cycle_fn(levels, Ps, coarse_scale=scale_c, eig_ratio=ratio, lam_mode=lam_mode)
            x, its, reason, rn, hist = O.pcg_with(prob.rowptr, prob.cols, prob.vals, prob.rhs, M, rtol=1e-5, maxits=3000)
            oc = sum(L.nnz for L in levels) / levels[0].nnz
            print(f"  {kind} omega={omega:.2f} scale={scale_c} ratio={ratio} lam={lam_mode}: its={its} reason={reason} levels={[L.shape[0] for L in levels]} "
                  f"op.complexity={oc:.2f} lam={['%.2f' % v for v in lam]} ({time.time()-t1:.1f}s)")


if __name__ == "__main__":
    main()
