"""The reference's own driver PROGRAMs (compiled in place by `make -C oracle drivers` against the
build-owned Fortran modules + libpfem_amd.so) run end to end: mesh files in, temp.dat out.

This is the "drop onto it unchanged" check of the boundary: the Fortran element loop calls
StiffnessResidual* / MatSetValues / VecSetValues once per element exactly as written in
tetrapoissonparallelimpl1.F:786-884, and solverpetsc%factoriseAndSolve runs the GPU solve.
The executables are reference-derived build products (oracle/_ref, never committed)."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import pfemfort_amd as pf
from oracle import pfem_oracle as O
from pfemfort_amd import host as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def _exe(name):
    path = os.path.join(REF, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (needs /root/reference + flang in the build container)")
    return path


def _write_mesh(mesh, prefix, ndof):
    with open(prefix + "-nodes.dat", "w") as f:
        for i in range(mesh.nNode):
            f.write("%d\t%.8f\t%.8f\t%.8f\n" % (i + 1, *mesh.xyz[:, i]))
    with open(prefix + "-elems.dat", "w") as f:
        for e in range(mesh.nElem):
            f.write("%d\t%d\t%d\t%d\t%d\n" % (e + 1, *(mesh.conn[:, e] + 1)))
    with open(prefix + "-DirichBC.dat", "w") as f:
        for n, d, v in zip(mesh.bc_node, mesh.bc_dof, mesh.bc_val):
            f.write("%d\t%d\t%.8f\n" % (n + 1, d + 1, v))


def _run(exe, prefix, cwd, rtol="1e-10"):
    env = dict(os.environ, PFEM_KSP_RTOL=rtol)
    return subprocess.run([exe, prefix + "-nodes.dat", prefix + "-elems.dat", prefix + "-DirichBC.dat"], cwd=cwd, env=env,
                          capture_output=True, text=True, timeout=600)


@pytest.mark.skipif(pf.device_count() > 0, reason="a GPU is present")
@pytest.mark.parametrize("nranks", [1, 2])
def test_unchanged_driver_links_runs_bookkeeping_and_fails_loudly_without_gpu(tmp_path, golden_dir, nranks):
    exe = _exe("tetrapoissonparallelimpl1" if nranks == 1 else "tetrapoissonparallelimpl1_mpi")
    for k in ("nodes", "elems", "DirichBC"):
        with gzip.open(os.path.join(golden_dir, "input", f"tet10-{k}.dat.gz"), "rb") as src, \
                open(tmp_path / f"tet10-{k}.dat", "wb") as dst:
            shutil.copyfileobj(src, dst)
    r = _run(exe, "tet10", tmp_path) if nranks == 1 else _run_mpi(exe, nranks, "tet10", tmp_path)
    out = r.stdout + r.stderr
    assert "Total DOF      =  729" in out               # the driver's own bookkeeping ran (:357-383)
    if nranks > 1:                                      # ... and its METIS call + MPI renumbering (:423-679)
        assert "After Metis" in out and "size_local" in out
    assert "no HIP device" in out                       # solverpetsc%initialise refuses: no CPU path
    assert nranks > 1 or r.returncode != 0              # (hydra does not forward a Fortran STOP code)


@pytest.mark.gpu
def test_unchanged_poisson_driver_on_tet10(tmp_path, golden_dir):
    exe = _exe("tetrapoissonparallelimpl1")
    for k in ("nodes", "elems", "DirichBC"):
        with gzip.open(os.path.join(golden_dir, "input", f"tet10-{k}.dat.gz"), "rb") as src, \
                open(tmp_path / f"tet10-{k}.dat", "wb") as dst:
            shutil.copyfileobj(src, dst)
    r = _run(exe, "tet10", tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Convergence in" in r.stdout and "Program is successful" in r.stdout
    t = np.loadtxt(tmp_path / "temp.dat")                # ii, old node, value  (:935-942)
    mesh = O.read_mesh(os.path.join(golden_dir, "input", "tet10"))
    prob = O.setup_problem(O.POISSON_TET, mesh)
    x, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assy = O.assy_for_soln(prob.dm.NodeDofArrayNew)
    assert np.array_equal(t[:, 0].astype(int), np.arange(1, 730))
    assert np.array_equal(t[:, 1].astype(int), prob.dm.node_map_get_old[assy] + 1)     # integer maps: bit-exact
    assert np.abs(t[:, 2] - x).max() < 1e-8
    exact = (mesh.xyz ** 2).sum(0)
    assert np.abs(t[:, 2] - exact[t[:, 1].astype(int) - 1]).max() < 2e-7
    assert os.path.exists(tmp_path / "Poisson-soln.vtk")  # the reference's own writervtk.F ran


@pytest.mark.gpu
def test_unchanged_elasticity_driver_on_small_beam(tmp_path):
    exe = _exe("tetraelasticityparallelimpl1")
    mesh = H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)
    _write_mesh(mesh, str(tmp_path / "beam"), 3)
    r = _run(exe, "beam", tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Convergence in" in r.stdout
    u = np.loadtxt(tmp_path / "temp.dat")                # one value per free dof (:1031-1046)
    prob = O.setup_problem(O.ELAST_TET, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val))
    x, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assert len(u) == len(x) and np.abs(u - x).max() < 1e-8 * max(1.0, np.abs(x).max())


# ---------------------------------------------------------------------------------------
# The same PROGRAMs under mpiexec: the reference's parallel path (METIS call, MPI renumbering,
# per-rank element loop with global indices, VecScatterCreateToAll) on the MPI flavour of the shim.
def _mpiexec():
    for cand in ("/opt/conda/bin/mpiexec", shutil.which("mpiexec")):
        if cand and os.path.exists(cand):
            return cand
    pytest.skip("mpiexec not available")


def _run_mpi(exe, nranks, prefix, cwd, rtol="1e-10"):
    env = dict(os.environ, PFEM_KSP_RTOL=rtol)
    return subprocess.run([_mpiexec(), "-n", str(nranks), exe, prefix + "-nodes.dat", prefix + "-elems.dat",
                           prefix + "-DirichBC.dat"], cwd=cwd, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3])
def test_unchanged_poisson_driver_under_mpiexec(tmp_path, golden_dir, nranks):
    exe = _exe("tetrapoissonparallelimpl1_mpi")
    for k in ("nodes", "elems", "DirichBC"):
        with gzip.open(os.path.join(golden_dir, "input", f"tet10-{k}.dat.gz"), "rb") as src, \
                open(tmp_path / f"tet10-{k}.dat", "wb") as dst:
            shutil.copyfileobj(src, dst)
    r = _run_mpi(exe, nranks, "tet10", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Convergence in" in r.stdout and "Program is successful" in r.stdout
    t = np.loadtxt(tmp_path / "temp.dat")                # written by rank 0: ii, old node, value
    mesh = O.read_mesh(os.path.join(golden_dir, "input", "tet10"))
    # the stand-in partitioner: contiguous node-index blocks (petsc_shim.f90: METIS_PartMeshNodal)
    npid = ((np.arange(mesh.nNode, dtype=np.int64) * nranks) // mesh.nNode).astype(np.int32)
    prob = O.setup_problem(O.POISSON_TET, mesh, nParts=nranks, node_proc_id=npid)
    assy = O.assy_for_soln(prob.dm.NodeDofArrayNew)
    assert np.array_equal(t[:, 0].astype(int), np.arange(1, 730))
    assert np.array_equal(t[:, 1].astype(int), prob.dm.node_map_get_old[assy] + 1)     # the driver's MPI renumbering
    x, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assert np.abs(t[:, 2] - x).max() < 1e-8
    exact = (mesh.xyz ** 2).sum(0)
    assert np.abs(t[:, 2] - exact[t[:, 1].astype(int) - 1]).max() < 2e-7


@pytest.mark.gpu
def test_unchanged_elasticity_driver_under_mpiexec(tmp_path):
    exe = _exe("tetraelasticityparallelimpl1_mpi")
    mesh = H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)
    _write_mesh(mesh, str(tmp_path / "beam"), 3)
    r = _run_mpi(exe, 2, "beam", tmp_path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Convergence in" in r.stdout
    u = np.loadtxt(tmp_path / "temp.dat")
    npid = ((np.arange(mesh.nNode, dtype=np.int64) * 2) // mesh.nNode).astype(np.int32)
    prob = O.setup_problem(O.ELAST_TET, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=2,
                           node_proc_id=npid)
    x, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assert len(u) == len(x) and np.abs(u - x).max() < 1e-8 * max(1.0, np.abs(x).max())


@pytest.mark.gpu
def test_mpi_driver_takes_a_metis_partition_from_files(tmp_path, golden_dir):
    """PFEM_METIS_PREFIX: epart/npart files in mpmetis' format replace the stand-in partitioner; an irregular
    3-part partition (angular sectors; interface nodes given to a pseudo-random adjacent part)."""
    exe = _exe("tetrapoissonparallelimpl1_mpi")
    for k in ("nodes", "elems", "DirichBC"):
        with gzip.open(os.path.join(golden_dir, "input", f"tet10-{k}.dat.gz"), "rb") as src, \
                open(tmp_path / f"tet10-{k}.dat", "wb") as dst:
            shutil.copyfileobj(src, dst)
    mesh = O.read_mesh(os.path.join(golden_dir, "input", "tet10"))
    world = 3
    cen = mesh.xyz[:, mesh.conn].mean(axis=1)
    epid = np.minimum(((np.arctan2(cen[1] + 0.013, cen[0] + 0.007) + np.pi) / (2 * np.pi) * world).astype(np.int32), world - 1)
    touch = np.zeros((world, mesh.nNode), bool)
    for a in range(4):
        touch[epid, mesh.conn[a]] = True
    npid = (np.random.default_rng(5).random((world, mesh.nNode)) * touch).argmax(axis=0).astype(np.int32)
    np.savetxt(tmp_path / "tet10.epart.3", epid, fmt="%d")
    np.savetxt(tmp_path / "tet10.npart.3", npid, fmt="%d")
    e2, n2 = H.read_metis_partition(str(tmp_path / "tet10"), 3)       # the Python side of the same hook
    assert np.array_equal(e2, epid) and np.array_equal(n2, npid)
    env = dict(os.environ, PFEM_KSP_RTOL="1e-10", PFEM_METIS_PREFIX="tet10")
    r = subprocess.run([_mpiexec(), "-n", "3", exe, "tet10-nodes.dat", "tet10-elems.dat", "tet10-DirichBC.dat"], cwd=tmp_path,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "Program is successful" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    t = np.loadtxt(tmp_path / "temp.dat")
    prob = O.setup_problem(O.POISSON_TET, mesh, nParts=world, node_proc_id=npid)
    assy = O.assy_for_soln(prob.dm.NodeDofArrayNew)
    assert np.array_equal(t[:, 1].astype(int), prob.dm.node_map_get_old[assy] + 1)
    x, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assert np.abs(t[:, 2] - x).max() < 1e-8


@pytest.mark.gpu
def test_elasticity_driver_with_pc_type_pbjacobi(tmp_path):
    """`-pc_type pbjacobi` in petsc_options.dat (what KSPSetFromOptions would read, tetraelasticityparallelimpl1.F:168)
    selects the node-block Jacobi: same solution in fewer iterations."""
    import re
    exe = _exe("tetraelasticityparallelimpl1")
    mesh = H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)
    _write_mesh(mesh, str(tmp_path / "beam"), 3)
    r0 = _run(exe, "beam", tmp_path)
    u0 = np.loadtxt(tmp_path / "temp.dat")
    (tmp_path / "petsc_options.dat").write_text("-ksp_type cg\n-pc_type pbjacobi\n")
    r1 = _run(exe, "beam", tmp_path)
    u1 = np.loadtxt(tmp_path / "temp.dat")
    assert r0.returncode == 0 and r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    its = [int(re.search(r"Convergence in\s+(\d+)", r.stdout).group(1)) for r in (r0, r1)]
    assert its[1] < its[0]
    assert np.abs(u1 - u0).max() < 1e-8 * max(1.0, np.abs(u0).max())
