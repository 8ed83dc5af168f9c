"""The reference's UNCHANGED driver programs, pinned as fixtures (tests/golden/drivers/*.npz, made in the build
container by tests/golden/make_driver_fixtures.py: the reference PROGRAMs with the reference's own element modules,
over the build-owned solver module): what THEY computed -- row blocks, ElemDofArray rows handed to MatSetValues,
lifted element vectors handed to VecSetValues, temp.dat -- against

  * the product's host bookkeeping and the oracle (CPU tests: integers bit-exact, lifted vectors bit-exact,
    solution <= 1e-10), and
  * the product's own driver harness on the GPU (``pfemfort_amd.drivers``: ``Result.temp_dat``), one rank and
    2-3 ranks sharing the device (GPU tests: integers bit-exact, values <= 1e-8).

No reference-derived binary is needed on the GPU box (tetrapoissonparallelimpl1.F:357-367, 500-679, 698-734,
828-884, 935-942; tetraelasticityparallelimpl1.F:1031-1050)."""
import glob
import gzip
import os
import shutil

import numpy as np
import pytest

from oracle import pfem_oracle as O
from pfemfort_amd import host as H

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "golden", "drivers", "*.npz")))


def _load(case):
    return np.load(os.path.join(HERE, "golden", "drivers", case + ".npz"))


def _mesh(case, tmp_path):
    if case.startswith("tet10"):
        for k in ("nodes", "elems", "DirichBC"):
            with gzip.open(os.path.join(HERE, "golden", "input", f"tet10-{k}.dat.gz"), "rb") as src, \
                    open(tmp_path / f"tet10-{k}.dat", "wb") as dst:
                shutil.copyfileobj(src, dst)
        return H.read_mesh(str(tmp_path / "tet10")), 1
    return H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3), 3


def _partition(fx, mesh):
    """(elem_proc_id, node_proc_id) the driver used: the METIS files of the fixture, else the shim's stand-in
    (contiguous node-index blocks; an element goes to the lowest part among its nodes)."""
    world = int(fx["nranks"])
    if "npart" in fx.files:
        return world, fx["epart"].astype(np.int32), fx["npart"].astype(np.int32)
    npid = ((np.arange(mesh.nNode, dtype=np.int64) * world) // mesh.nNode).astype(np.int32)
    return world, npid[mesh.conn].min(axis=0).astype(np.int32), npid


def test_fixtures_are_present():
    assert len(CASES) >= 7 and any("np3_metis" in c for c in CASES)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("side", ["product", "oracle"])
def test_driver_bookkeeping_matches_the_reference_run(case, side, tmp_path):
    fx = _load(case)
    mesh, ndof = _mesh(case, tmp_path)
    kind = O.POISSON_TET if ndof == 1 else O.ELAST_TET
    world, epid, npid = _partition(fx, mesh)
    M = H if side == "product" else O
    dm = M.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, npid)
    assert int(fx["size_global"]) == dm.size_global
    conn_new = dm.node_map_get_new[mesh.conn].astype(np.int32)
    xyz_new = np.ascontiguousarray(mesh.xyz[:, dm.node_map_get_old])
    assy = M.assy_for_soln(dm.NodeDofArrayNew)
    temp = fx["temp"]
    if ndof == 1:        # ii, OLD node of the ii-th free dof, value (:935-942)
        assert np.array_equal(temp[:, 0].astype(np.int64), np.arange(1, dm.size_global + 1))
        assert np.array_equal(temp[:, 1].astype(np.int64), dm.node_map_get_old[assy] + 1)
    ed = O.ELAST_ELEMDATA if ndof == 3 else O.POISSON_ELEMDATA
    for r in range(world):
        mine = np.nonzero(epid == r)[0]                                       # elem_proc_id(ee) == this_mpi_proc (:829)
        edof = M.elem_dof_array(np.ascontiguousarray(conn_new[:, mine]), dm.NodeDofArrayNew)
        assert int(fx[f"r{r}_size_local"]) == int(dm.row_end[r]) - int(dm.row_start[r])
        assert int(fx[f"r{r}_row_start"]) == int(dm.row_start[r])
        for key in ("insert_idx", "add_idx", "vec_idx"):                      # ElemDofArray rows: bit-exact
            assert np.array_equal(fx[f"r{r}_{key}"], edof.T)
        if side == "oracle" and len(mine):
            # the lifted element vectors the reference handed to VecSetValues (:859-880), from the oracle's Ke/Fe
            K, F = O.eval_elems(kind, xyz_new, np.ascontiguousarray(conn_new[:, mine]), ed)
            nsize = edof.shape[0]
            for ii in range(nsize):
                dirich = edof[ii] == -1
                fact = dm.solnApplied[conn_new[ii // ndof, mine] * ndof + ii % ndof]
                for jj in range(nsize):
                    upd = dirich & (edof[jj] != -1)
                    F[upd, jj] = F[upd, jj] - K[upd, jj, ii] * fact[upd]
            assert np.array_equal(fx[f"r{r}_vec_val"], F)
    if side == "oracle":
        prob = O.setup_problem(kind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=world,
                               node_proc_id=npid)
        x, its, reason, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
        vals = temp[:, 2] if ndof == 1 else temp
        assert reason == 2 and abs(its - int(fx["its"])) <= 3       # entries summed in another order (rank by rank)
        assert np.abs(vals - x).max() <= 1e-10 * max(1.0, np.abs(x).max())


# ---------------------------------------------------------------------------------------
def _gpu_rank(rank, world, port, case, out_dir, pc=None):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathlib
        import pfemfort_amd as pf
        from pfemfort_amd import drivers as D
        fx = _load(case)
        rank_dir = pathlib.Path(out_dir) / f"rank{rank}"          # every rank unpacks its own copy of the input files
        rank_dir.mkdir()
        mesh, ndof = _mesh(case, rank_dir)
        kind = pf.POISSON_TET if ndof == 1 else pf.ELAST_TET
        _, epid, npid = _partition(fx, mesh)
        res = D.run_parallel(kind, mesh, epid, npid, dist, torch, rtol=1e-12, staged=True, pc=pc)
        if rank == 0:
            res.write_outputs(out_dir)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("pc", [None, "gamg"])
@pytest.mark.parametrize("case", CASES)
def test_gpu_driver_harness_reproduces_temp_dat(case, pc, tmp_path):
    """The build's own counterpart of the driver (pfemfort_amd.drivers) on the GPU writes the same temp.dat as the
    reference program did: integer columns bit-exact, values <= 1e-8 -- with the default point Jacobi and with
    -pc_type gamg (on several ranks: ONE multigrid hierarchy across the ranks, the library's multi-rank default)."""
    import pfemfort_amd as pf
    fx = _load(case)
    world = int(fx["nranks"])
    if world == 1:
        mesh, ndof = _mesh(case, tmp_path)
        drv = pf.tetrapoissonparallelimpl1 if ndof == 1 else pf.tetraelasticityparallelimpl1
        drv(mesh, rtol=1e-12, pc=pc).write_outputs(str(tmp_path))
    else:
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        mp.spawn(_gpu_rank, args=(world, port, case, str(tmp_path), pc), nprocs=world, join=True)
    got = np.loadtxt(tmp_path / "temp.dat")
    want = fx["temp"]
    assert got.shape == want.shape
    if want.ndim == 2:
        assert np.array_equal(got[:, :2], want[:, :2])
        got, want = got[:, 2], want[:, 2]
    assert np.abs(got - want).max() <= 1e-8 * max(1.0, np.abs(want).max())


# ---------------------------------------------------------------------------------------
def _gpu_rank_2d(rank, world, port, which, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pfemfort_amd as pf
        from pfemfort_amd import drivers as D
        mesh, kind, ed = _mesh_2d(which, pf)
        epid, npid = _sectors_2d(mesh, world)
        res = D.run_parallel(kind, mesh, epid, npid, dist, torch, elemData=ed, rtol=1e-12, maxits=100000, staged=True)
        if rank == 0:
            np.save(os.path.join(out_dir, "u.npy"), res.soln_free)
    finally:
        dist.destroy_process_group()


def _mesh_2d(which, K):
    if which == "tria20_poisson":
        return H.read_mesh(os.path.join(HERE, "golden", "input", "tria20x20")), K.POISSON_TRIA, np.array([1.0, 1.0])
    return H.read_mesh(os.path.join(HERE, "golden", "input", "cookmembranetria32")), K.ELAST_TRIA, H.ELAST2D_ELEMDATA


def _sectors_2d(mesh, world):
    cen = mesh.xyz[:, mesh.conn].mean(axis=1)
    ang = np.arctan2(cen[1] - cen[1].mean() + 0.013, cen[0] - cen[0].mean() + 0.007)
    epid = np.minimum(((ang + np.pi) / (2 * np.pi) * world).astype(np.int32), world - 1)
    touch = np.zeros((world, mesh.nNode), bool)
    for a in range(mesh.conn.shape[0]):
        touch[epid, mesh.conn[a]] = True
    npid = (np.random.default_rng(3).random((world, mesh.nNode)) * touch).argmax(axis=0).astype(np.int32)
    return epid, npid


@pytest.mark.gpu
@pytest.mark.parametrize("which,world", [("tria20_poisson", 2), ("cook_elasticity", 3)])
def test_gpu_two_dimensional_siblings_on_several_ranks(which, world, tmp_path):
    """The 2-D siblings of the path (triapoissonparallelimpl1 / triaelasticityparallelimpl1) through the same
    multi-rank harness: irregular sector partition, nodal forces on Cook's membrane, against a direct solve of the
    oracle-assembled system."""
    import socket
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_gpu_rank_2d, args=(world, port, which, str(tmp_path)), nprocs=world, join=True)
    mesh, kind, ed = _mesh_2d(which, O)
    _, npid = _sectors_2d(mesh, world)
    prob = O.setup_problem(kind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), elemData=ed, nParts=world,
                           node_proc_id=npid)
    rhs = prob.rhs.copy()
    if mesh.force_node is not None and which.startswith("cook"):
        rhs[prob.dm.NodeDofArrayNew[prob.dm.node_map_get_new[mesh.force_node], mesh.force_dof]] += mesh.force_val
    u = spl.spsolve(sp.csr_matrix((prob.vals, prob.cols, prob.rowptr)).tocsc(), rhs)
    got = np.load(tmp_path / "u.npy")
    assert np.abs(got - u).max() <= 1e-8 * max(1.0, np.abs(u).max())
