"""The N>1 path, world_size 2, 127.0.0.1 rendezvous.

CPU (gloo): pins the product's host logic for the sub-assembled layout -- slab partition,
reference renumbering, ghost detection, interface plan, the torch.distributed all-reduce hook --
by running a numpy restatement of the device CG loop on each rank (local operators from the
oracle) and comparing with the single-rank solution.

GPU (2-3 ranks on cuda:0, exchange staged through the host and reduced by gloo on CPU tensors, as the MPI
flavour does): the same layout through the real C++/HIP path (pfem_solver_set_comm / _set_interface / run_pcg
with the hook), against the oracle's direct solve and iteration count.
"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _partition(mesh, world, how, H):
    """(elem_proc_id, node_proc_id): z-slabs, or an irregular METIS-like one -- elements by the
    angular sector of their centroid, every node given to a pseudo-random part among those of the
    elements touching it (so interface nodes are shared by up to `world` ranks and the owned row
    blocks interleave in space).  Deterministic: every rank computes the same arrays."""
    if how == "slabs":
        return H.partition_box_slabs(*mesh.box, world)
    cen = mesh.xyz[:, mesh.conn].mean(axis=1)
    ang = np.arctan2(cen[1] - cen[1].mean() + 0.013, cen[0] - cen[0].mean() + 0.007)
    epid = np.minimum(((ang + np.pi) / (2 * np.pi) * world).astype(np.int32), world - 1)
    touch = np.zeros((world, mesh.nNode), bool)
    for a in range(mesh.conn.shape[0]):
        touch[epid, mesh.conn[a]] = True
    rng = np.random.default_rng(12345)
    pick = rng.random((world, mesh.nNode)) * touch
    return epid, pick.argmax(axis=0).astype(np.int32)


def _rank_setup(rank, world, kind, mesh_args, H, PD, dist):
    """Everything a rank does before the element loop (bench.py does the same)."""
    mesh = H.gen_box_tets(*mesh_args["box"], bc_mode=mesh_args["bc_mode"], ndof=mesh_args["ndof"])
    epid, npid = _partition(mesh, world, mesh_args.get("partition", "slabs"), H)
    ndof = mesh_args["ndof"]
    dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, npid)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)                         # :659-664, :832-838
    mine = np.nonzero(epid == rank)[0]                       # elem_proc_id(ee)==this_mpi_proc (:829)
    conn_loc = np.ascontiguousarray(conn_new[:, mine])
    edof_loc = H.elem_dof_array(conn_loc, dm.NodeDofArrayNew)
    rs, re = int(dm.row_start[rank]), int(dm.row_end[rank])
    return mesh, dm, conn_loc, xyz_new, edof_loc, rs, re


def _cpu_worker(rank, world, port, kind, mesh_args, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pfem_oracle as O
        from pfemfort_amd import distributed as PD
        from pfemfort_amd import host as H
        mesh, dm, conn_loc, xyz_new, edof_g, rs, re = _rank_setup(rank, world, kind, mesh_args, H, PD, dist)
        n_owned = re - rs
        ghosts = H.find_ghosts(edof_g, rs, n_owned)
        lists = PD.gather_ghost_lists(ghosts, dist)
        ranges = [None] * world
        dist.all_gather_object(ranges, (rs, re))
        gid, slot, n_iface = PD.interface_plan(lists, ranges, rank)
        # local numbering exactly as the device does it (k_localize_dofs): owned first, ghosts after
        n_loc = n_owned + len(ghosts)
        e = edof_g.astype(np.int64)
        loc = np.where((e >= rs) & (e < re), e - rs, n_owned + np.searchsorted(ghosts, e))
        edof_l = np.where(e < 0, -1, loc).astype(np.int32)
        lidx = np.where((gid >= rs) & (gid < re), gid - rs, n_owned + np.searchsorted(ghosts, gid))
        # local sub-assembled operator and rhs from the oracle (this rank's elements only)
        ed = O.ELAST_ELEMDATA if kind == O.ELAST_TET else O.POISSON_ELEMDATA
        rowptr, cols = O.csr_pattern(edof_l, n_loc)
        vals, rhs = O.assemble(kind, xyz_new, conn_loc, edof_l, dm.solnApplied, ed, n_loc, rowptr, cols)

        xbuf = torch.zeros(n_iface + 4, dtype=torch.float64)
        hook = PD.TorchAllReduce(dist, xbuf)
        xb = xbuf.numpy()

        def iface_sum(v, extra=()):
            xb[:n_iface + len(extra)] = 0.0
            xb[slot] = v[lidx]
            for j, s in enumerate(extra):
                xb[n_iface + j] = s
            assert hook(None, hook.base, n_iface + len(extra), None) == 0
            v[lidx] = xb[slot]
            return [xb[n_iface + j] for j in range(len(extra))]

        def sum2(a, b):
            xb[n_iface + 2:n_iface + 4] = (a, b)
            assert hook(None, hook.base + 8 * (n_iface + 2), 2, None) == 0
            return xb[n_iface + 2], xb[n_iface + 3]

        # run_pcg (csrc/pfem_device.hip) restated in numpy
        diag = np.array([vals[rowptr[i]:rowptr[i + 1]][cols[rowptr[i]:rowptr[i + 1]] == i].sum() for i in range(n_loc)])
        iface_sum(diag)
        iface_sum(rhs)
        dinv = 1.0 / diag
        x = np.zeros(n_loc); r = rhs.copy(); p = r * dinv
        own = slice(0, n_owned)
        beta, zz = sum2(float(r[own] @ p[own]), float(p[own] @ p[own]))
        rn0 = np.sqrt(zz); ttol = max(1e-10 * rn0, 1e-50)
        its = 0
        for it in range(5000):
            w = O.spmv(rowptr, cols, vals, p)
            (pw,) = iface_sum(w, extra=(float(p @ w),))           # (p,A_loc p) over ALL local rows
            alpha = beta / pw
            x += alpha * p; r -= alpha * w
            z = r * dinv
            rz, zz = sum2(float(r[own] @ z[own]), float(z[own] @ z[own]))
            its = it + 1
            if np.sqrt(zz) <= ttol:
                break
            p = z + (rz / beta) * p
            beta = rz
        assert hook.error is None
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x[own], rs=rs, re=re, its=its, n_iface=n_iface,
                 n_ghost=len(ghosts), calls=hook.calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind_name,world,partition", [("poisson", 2, "slabs"), ("elast", 2, "slabs"),
                                                       ("poisson", 3, "sectors")])
def test_gloo_subassembled_cg_matches_serial(tmp_path, kind_name, world, partition):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch.multiprocessing as mp
    from oracle import pfem_oracle as O
    kind = O.POISSON_TET if kind_name == "poisson" else O.ELAST_TET
    mesh_args = ({"box": (-1, 1, 6, -1, 1, 5, -1, 1, 7), "bc_mode": 0, "ndof": 1} if kind_name == "poisson" else
                 {"box": (-0.5, 0.5, 2, 0.0, 3.0, 6, -0.5, 0.5, 4), "bc_mode": 1, "ndof": 3})
    mesh_args["partition"] = partition
    mp.spawn(_cpu_worker, args=(world, _free_port(), kind, mesh_args, str(tmp_path)), nprocs=world, join=True)
    # serial truth: oracle assembly + direct solve on the SAME (renumbered) global problem
    from pfemfort_amd import host as H
    mesh = H.gen_box_tets(*mesh_args["box"], bc_mode=mesh_args["bc_mode"], ndof=mesh_args["ndof"])
    _, npid = _partition(mesh, world, partition, H)
    prob = O.setup_problem(kind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=world,
                           node_proc_id=npid)
    u = spl.spsolve(sp.csr_matrix((prob.vals, prob.cols, prob.rowptr)).tocsc(), prob.rhs)
    got = np.empty_like(u)
    tot = 0
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        got[int(d["rs"]):int(d["re"])] = d["x"]
        tot += int(d["re"]) - int(d["rs"])
        assert int(d["n_iface"]) > 0 and int(d["calls"]) >= 2 * int(d["its"])
    assert tot == len(u)
    assert np.abs(got - u).max() <= 1e-8 * max(1.0, np.abs(u).max())


def test_interface_plan_small_example():
    from pfemfort_amd import distributed as PD
    # rank0 owns [0,5), rank1 [5,9), rank2 [9,12); ghosts are what each touches but does not own
    lists = [np.array([5, 6]), np.array([3, 4, 9]), np.array([6, 8])]
    ranges = [(0, 5), (5, 9), (9, 12)]
    iface = [3, 4, 5, 6, 8, 9]
    for r, (exp_g) in enumerate([[3, 4, 5, 6], [3, 4, 5, 6, 8, 9], [6, 8, 9]]):
        gid, slot, n = PD.interface_plan(lists, ranges, r)
        assert n == 6 and gid.tolist() == exp_g and [iface[s] for s in slot] == exp_g


# ---------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, mesh_args, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)     # a stuck rank reports where, instead of hanging the suite
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pfemfort_amd as pf
        from pfemfort_amd import distributed as PD
        from pfemfort_amd import host as H
        kind = pf.POISSON_TET if mesh_args["ndof"] == 1 else pf.ELAST_TET
        mesh, dm, conn_loc, xyz_new, edof_g, rs, re = _rank_setup(rank, world, kind, mesh_args, H, PD, dist)
        s = pf.PetscSolver().initialise(re - rs, dm.size_global, row_start=rs, device=0)
        s.setTolerances(rtol=1e-10)
        if mesh_args.get("pc"):
            s.setPreconditioner(mesh_args["pc"])
        s.setSpmvFormat(mesh_args.get("spmv", "auto"))
        ed = H.ELAST_ELEMDATA if kind == pf.ELAST_TET else H.POISSON_ELEMDATA
        if mesh_args.get("mode", "batched") == "batched":
            s.uploadMesh(kind, conn_loc, xyz_new, edof_g, dm.solnApplied)
            hook, n_iface = PD.attach(s, dist, torch, torch.device("cuda", 0), staged=True)
            s.buildPattern()
            s.assemble(ed, H.TIMEDATA)
        else:
            # the Fortran driver's own loops (tetrapoissonparallelimpl1.F:786-884) with GLOBAL indices: pattern by
            # INSERT_VALUES, setZero, then per-element routine + lifting + ADD_VALUES for this rank's elements
            from pfemfort_amd.solver import ADD_VALUES, INSERT_VALUES
            ndof, nsize = mesh_args["ndof"], edof_g.shape[0]
            for e in range(conn_loc.shape[1]):
                s.MatSetValues(edof_g[:, e], edof_g[:, e], np.zeros(nsize * nsize), INSERT_VALUES)
            s.setZero()
            hook, n_iface = PD.attach(s, dist, torch, torch.device("cuda", 0), staged=True)
            fn = H.StiffnessResidualElasticityLinearTetra if kind == pf.ELAST_TET else H.StiffnessResidualPoissonLinearTetra
            for e in range(conn_loc.shape[1]):
                nd = conn_loc[:, e]
                K, F = fn(xyz_new[0, nd], xyz_new[1, nd], xyz_new[2, nd], ed, H.TIMEDATA, np.zeros(nsize))
                f = edof_g[:, e]
                s.MatSetValues(f, f, K.ravel(order="F"), ADD_VALUES)
                for ii in np.nonzero(f == -1)[0]:
                    fact = dm.solnApplied[nd[ii // ndof] * ndof + ii % ndof]
                    F = F - np.where(f != -1, K[:, ii] * fact, 0.0)
                s.VecSetValues(f, F, ADD_VALUES)
        hook.log = []
        its, reason, rn = s.factoriseAndSolve()
        assert hook.error is None, hook.error
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=s.getSolution(), rs=rs, re=re, its=its, reason=reason,
                 n_iface=n_iface, pc=s.preconditioner(), calls=hook.calls, log=np.array([c for _, c in hook.log]))
        s.free()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind_name,world,partition,mode", [("poisson", 2, "slabs", "batched"), ("elast", 2, "slabs", "batched"),
                                                            ("poisson", 3, "sectors", "batched"), ("elast", 3, "sectors", "batched"),
                                                            ("poisson", 2, "slabs", "compat"), ("elast", 3, "sectors", "compat"),
                                                            ("elast", 2, "slabs", "pbjacobi"), ("elast", 3, "sectors", "pbjacobi"),
                                                            ("poisson", 2, "slabs", "pbjacobi")])
def test_gpu_ranks_on_one_device_match_oracle(tmp_path, kind_name, world, partition, mode):
    """2-3 ranks share cuda:0 (host-staged gloo exchange): the product's multi-rank device loop against the ORACLE --
    a direct solve of the oracle-assembled global system in the partition's new numbering, and the oracle's
    Jacobi-PCG iteration count."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch.multiprocessing as mp
    from oracle import pfem_oracle as O
    from pfemfort_amd import host as H
    mesh_args = ({"box": (-1, 1, 12, -1, 1, 10, -1, 1, 14), "bc_mode": 0, "ndof": 1} if kind_name == "poisson" else
                 {"box": (-0.5, 0.5, 3, 0.0, 3.0, 8, -0.5, 0.5, 6), "bc_mode": 1, "ndof": 3})
    mesh_args["partition"] = partition
    mesh_args["mode"] = mode
    if world == 3 or mode != "batched":   # the row-group SpMV forms ("auto" keeps systems this small in the row form)
        mesh_args["spmv"] = "grouped"
    if mode == "pbjacobi":            # node-block Jacobi on several ranks (blocks of shared nodes summed, groups voted)
        mesh_args["mode"], mesh_args["pc"] = "batched", "pbjacobi"
    if mode == "compat" and kind_name == "poisson":
        mesh_args["box"] = (-1, 1, 6, -1, 1, 5, -1, 1, 7)
    mp.spawn(_gpu_worker, args=(world, _free_port(), mesh_args, str(tmp_path)), nprocs=world, join=True)
    mesh = H.gen_box_tets(*mesh_args["box"], bc_mode=mesh_args["bc_mode"], ndof=mesh_args["ndof"])
    kind = O.POISSON_TET if kind_name == "poisson" else O.ELAST_TET
    _, npid = _partition(mesh, world, partition, H)
    prob = O.setup_problem(kind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=world,
                           node_proc_id=npid)
    u = spl.spsolve(sp.csr_matrix((prob.vals, prob.cols, prob.rowptr)).tocsc(), prob.rhs)
    _, its_oracle, reason_oracle, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-10)
    assert reason_oracle == 2
    its_tol = 3
    if mode == "pbjacobi":
        want = "pbjacobi" if kind_name == "elast" else "jacobi"      # Poisson has no multi-row groups: all ranks fall back
        assert all(str(np.load(tmp_path / f"rank{r}.npz")["pc"]) == want for r in range(world))
        if kind_name == "elast":
            d0 = np.load(tmp_path / "rank0.npz")
            assert int(d0["its"]) < its_oracle                       # fewer iterations than point Jacobi ...
            its_tol = 10 ** 9                                        # ... so the count is not compared below
    got = np.full_like(u, np.nan)
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert (int(d["rs"]), int(d["re"])) == (int(prob.dm.row_start[r]), int(prob.dm.row_end[r]))
        got[int(d["rs"]):int(d["re"])] = d["x"]
        assert int(d["reason"]) == 2 and abs(int(d["its"]) - its_oracle) <= its_tol
        assert int(d["calls"]) >= 2 * int(d["its"])
        # every rank issued the same sequence of exchanges (count per call)
        assert np.array_equal(d["log"], np.load(tmp_path / "rank0.npz")["log"])
    assert np.abs(got - u).max() <= 1e-8 * max(1.0, np.abs(u).max())


def _nccl_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import pfemfort_amd as pf
        from pfemfort_amd import distributed as PD
        from pfemfort_amd import host as H
        mesh = H.gen_box_tets(-1, 1, 6, -1, 1, 5, -1, 1, 7)
        dm = H.dof_numbering(mesh.nNode, 1, mesh.bc_node, mesh.bc_dof, mesh.bc_val)
        conn, xyz = H.renumber_mesh(mesh, dm)
        edof = H.elem_dof_array(conn, dm.NodeDofArrayNew)
        s = pf.PetscSolver().initialise(dm.size_global, dm.size_global, device=0)
        s.setTolerances(rtol=1e-10)
        s.uploadMesh(pf.POISSON_TET, conn, xyz, edof, dm.solnApplied)
        hook, n_iface = PD.attach(s, dist, torch, torch.device("cuda", 0))      # all_gather_object over RCCL
        s.buildPattern()
        s.assemble(H.POISSON_ELEMDATA, H.TIMEDATA)
        its, reason, _ = s.factoriseAndSolve()
        # the hook itself, as the library calls it: in-place SUM on a slice of the exchange tensor
        hook.xbuf[:] = torch.arange(hook.xbuf.numel(), dtype=torch.float64, device="cuda")
        assert hook(None, hook.base + 8, 3, None) == 0 and hook.error is None
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "nccl.npz"), its=its, reason=reason, n_iface=n_iface, x=s.getSolution(),
                 xbuf=hook.xbuf.cpu().numpy())
        s.free()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_backend_binds_to_the_hook_world_size_1(tmp_path):
    """Only one GPU per box here, and RCCL refuses two ranks on one device: this pins what CAN be pinned of the
    backend bench.py uses for N>1 -- process-group creation on the device, the object all-gather of attach(), and an
    in-place float64 all_reduce on a slice of the exchange tensor issued through the hook -- with world_size 1."""
    import torch.multiprocessing as mp
    import pfemfort_amd as pf
    from pfemfort_amd import host as H
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d = np.load(tmp_path / "nccl.npz")
    ref = pf.tetrapoissonparallelimpl1(H.gen_box_tets(-1, 1, 6, -1, 1, 5, -1, 1, 7), rtol=1e-10)
    assert int(d["reason"]) == 2 and int(d["its"]) == ref.its and int(d["n_iface"]) == 0
    assert np.array_equal(d["x"], ref.soln_free)
    assert np.array_equal(d["xbuf"], np.arange(len(d["xbuf"]), dtype=np.float64))   # SUM over one rank: unchanged
