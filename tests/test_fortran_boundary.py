"""The Fortran side of the drop-in boundary on the GPU, through the build's OWN Fortran host program
(tests/native/boundary_check.F90, built by `make -C pfemfort_amd/fortran check`): the PETSc finclude files and modules,
TYPE PetscSolver of Module_SolverPetsc, the element modules, MatSetValues / VecSetValues / VecScatterCreateToAll /
the legacy VecGetArray -- the call surface tetrapoissonparallelimpl1.F / tetraelasticityparallelimpl1.F use --
one process, and under mpiexec with 2-3 ranks sharing the GPU (pfem_mpi.cpp: neighbour plan + MPI host hooks).
The program carries no mesh bookkeeping; the prepared problem comes from the product's host routines, and the
answer is checked against the oracle and against the fixtures of the reference's own driver runs
(tests/golden/drivers).  No reference-derived binary is involved."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import pfemfort_amd as pf
from oracle import pfem_oracle as O
from pfemfort_amd import host as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORT = os.path.join(ROOT, "pfemfort_amd", "fortran")


def _exe(mpi):
    path = os.path.join(FORT, "build_mpi", "boundary_check_mpi") if mpi else os.path.join(FORT, "build", "boundary_check")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (flang / MPI not present where the tree was built)")
    return path


def _mpiexec():
    for cand in ("/opt/conda/bin/mpiexec", shutil.which("mpiexec")):
        if cand and os.path.exists(cand):
            return cand
    pytest.skip("mpiexec not available")


def _prepare(mesh, ndof, world, path):
    """problem.txt for boundary_check: the new numbering, ElemDofArray and ownership from the product's host routines
    (partition = the shim's stand-in: contiguous node-index blocks, an element goes to the lowest part of its nodes)."""
    npid = ((np.arange(mesh.nNode, dtype=np.int64) * world) // mesh.nNode).astype(np.int32)
    epid = npid[mesh.conn].min(axis=0).astype(np.int32)
    dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, npid)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)
    edof = H.elem_dof_array(conn_new, dm.NodeDofArrayNew)
    ed = np.zeros(6)
    src = H.ELAST_ELEMDATA if ndof == 3 else H.POISSON_ELEMDATA
    ed[:len(src)] = src
    with open(path, "w") as f:
        f.write(f"{ndof} {mesh.nNode} {mesh.nElem} {dm.size_global} {world}\n")
        f.write(" ".join(str(int(dm.row_end[r] - dm.row_start[r])) for r in range(world)) + "\n")
        np.savetxt(f, epid[None, :], fmt="%d")
        np.savetxt(f, xyz_new.T, fmt="%.17g")                       # Fortran reads xyz(3, nNode) column by column
        np.savetxt(f, conn_new.T + 1, fmt="%d")
        np.savetxt(f, edof.T, fmt="%d")
        np.savetxt(f, dm.solnApplied[None, :], fmt="%.17g")
        np.savetxt(f, ed[None, :], fmt="%.17g")
    return dm, npid


def _run(exe, cwd, world, rtol="1e-10", env_extra=None):
    env = dict(os.environ, PFEM_KSP_RTOL=rtol, **(env_extra or {}))
    cmd = [exe] if world == 1 else [_mpiexec(), "-n", str(world), exe]
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    return r


@pytest.mark.skipif(pf.device_count() > 0, reason="a GPU is present")
def test_boundary_program_fails_loudly_without_gpu(tmp_path):
    exe = _exe(False)
    mesh = H.gen_box_tets(-1, 1, 2, -1, 1, 2, -1, 1, 2)
    _prepare(mesh, 1, 1, tmp_path / "problem.txt")
    r = _run(exe, tmp_path, 1)
    assert r.returncode != 0 and "no HIP device" in (r.stdout + r.stderr)
    assert "PC = jacobi" in r.stdout                                 # the one loud line about the preconditioner


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host_loop", "device"])
@pytest.mark.parametrize("case,world", [("tet10", 1), ("tet10", 2), ("tet10", 3), ("beam", 1), ("beam", 2)])
def test_fortran_host_program_on_gpu(tmp_path, golden_dir, case, world, mode):
    """mode host_loop: the driver's own element loop (element routine + MatSetValues / VecSetValues per element), the GPU
    solves.  mode device: PetscSolver%uploadMeshToDevice + %assembleOnDevice -- the element loop itself runs in the HIP
    kernels, called from Fortran."""
    exe = _exe(world > 1)
    if case == "tet10":
        mesh, ndof, kind = H.read_mesh(f"{golden_dir}/input/tet10"), 1, O.POISSON_TET
        fixture = f"tet10_poisson_np{world}"
    else:
        mesh, ndof, kind = H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3), 3, O.ELAST_TET
        fixture = f"beam3x12x3_elast_np{world}"
    dm, npid = _prepare(mesh, ndof, world, tmp_path / "problem.txt")
    r = _run(exe, tmp_path, world, rtol="1e-12", env_extra={"PFEM_CHECK_MODE": mode})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Convergence in" in r.stdout and "PC = jacobi" in r.stdout
    lines = open(tmp_path / "solution.txt").read().split()
    its, reason = int(lines[0]), int(lines[1])
    u = np.array(lines[2:], dtype=float)
    prob = O.setup_problem(kind, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=world,
                           node_proc_id=npid)
    x, its_o, reason_o, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-12)
    assert reason == reason_o == 2 and abs(its - its_o) <= 3 and len(u) == len(x)
    assert np.abs(u - x).max() <= 1e-8 * max(1.0, np.abs(x).max())
    # ... and the values the reference's own driver wrote for the same case (temp.dat of the fixture)
    fx = np.load(os.path.join(golden_dir, "drivers", fixture + ".npz"))
    want = fx["temp"][:, 2] if ndof == 1 else fx["temp"]
    assert np.abs(u - want).max() <= 1e-8 * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_fortran_host_program_pc_type_pbjacobi_and_unknown_options(tmp_path):
    """`-pc_type pbjacobi` in petsc_options.dat (what KSPSetFromOptions would read) selects the node-block Jacobi: same
    solution in fewer iterations; an unknown -pc_type stops loudly instead of silently running another solve."""
    exe = _exe(False)
    mesh = H.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)
    _prepare(mesh, 3, 1, tmp_path / "problem.txt")
    r0 = _run(exe, tmp_path, 1)
    u0 = np.array(open(tmp_path / "solution.txt").read().split(), dtype=float)
    (tmp_path / "petsc_options.dat").write_text("-ksp_type cg\n-pc_type pbjacobi\n")
    r1 = _run(exe, tmp_path, 1)
    u1 = np.array(open(tmp_path / "solution.txt").read().split(), dtype=float)
    assert r0.returncode == 0 and r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    assert "PC = pbjacobi" in r1.stdout and u1[0] < u0[0]                 # fewer iterations
    assert np.abs(u1[2:] - u0[2:]).max() < 1e-8 * max(1.0, np.abs(u0[2:]).max())
    # -pc_type gamg from the options file: the multigrid V-cycle, from the Fortran host program, far fewer iterations
    (tmp_path / "petsc_options.dat").write_text("-ksp_type cg\n-pc_type gamg\n")
    r3 = _run(exe, tmp_path, 1)
    u3 = np.array(open(tmp_path / "solution.txt").read().split(), dtype=float)
    assert r3.returncode == 0 and "PC = gamg" in r3.stdout and u3[1] == 2 and u3[0] < 0.5 * u1[0], r3.stdout[-2000:] + r3.stderr[-2000:]
    assert np.abs(u3[2:] - u0[2:]).max() < 1e-8 * max(1.0, np.abs(u0[2:]).max())
    # -pc_mg_cycle_type w (PCMGSetCycleType): accepted, same solution; anything but v / w stops loudly
    (tmp_path / "petsc_options.dat").write_text("-ksp_type cg\n-pc_type gamg\n-pc_mg_cycle_type w\n")
    r4 = _run(exe, tmp_path, 1)
    u4 = np.array(open(tmp_path / "solution.txt").read().split(), dtype=float)
    assert r4.returncode == 0 and u4[1] == 2 and u4[0] <= u3[0], r4.stdout[-2000:] + r4.stderr[-2000:]
    assert np.abs(u4[2:] - u0[2:]).max() < 1e-8 * max(1.0, np.abs(u0[2:]).max())
    (tmp_path / "petsc_options.dat").write_text("-pc_type gamg\n-pc_mg_cycle_type f\n")
    r5 = _run(exe, tmp_path, 1)
    assert r5.returncode != 0 and "-pc_mg_cycle_type f is not available" in (r5.stdout + r5.stderr).replace("  ", " ")
    (tmp_path / "petsc_options.dat").write_text("-pc_type ilu\n")
    r2 = _run(exe, tmp_path, 1)
    assert r2.returncode != 0 and "-pc_type ilu is not available" in (r2.stdout + r2.stderr).replace("  ", " ")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_fortran_host_program_under_mpiexec_with_pc_type_gamg(tmp_path, golden_dir, world):
    """`-pc_type gamg` from the options file with the Fortran host program under mpiexec: the ranks (sharing the GPU, MPI
    host hooks of pfem_mpi.cpp) form ONE multigrid hierarchy across them on the reference-style partition of the tet10
    mesh (node-index blocks: interface dofs with several holders); same solution as the oracle's, about the
    iterations of the one-rank hierarchy, far fewer than point Jacobi."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    mesh = H.read_mesh(f"{golden_dir}/input/tet10")
    dm, npid = _prepare(mesh, 1, world, tmp_path / "problem.txt")
    (tmp_path / "petsc_options.dat").write_text("-ksp_type cg\n-pc_type gamg\n")
    r = _run(_exe(True), tmp_path, world, rtol="1e-10", env_extra={"PFEM_CHECK_MODE": "device"})
    assert r.returncode == 0 and "PC = gamg" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    lines = open(tmp_path / "solution.txt").read().split()
    its, reason = int(lines[0]), int(lines[1])
    u = np.array(lines[2:], dtype=float)
    prob = O.setup_problem(O.POISSON_TET, O.Mesh(mesh.xyz, mesh.conn, mesh.bc_node, mesh.bc_dof, mesh.bc_val), nParts=world, node_proc_id=npid)
    x = spl.spsolve(sp.csr_matrix((prob.vals, prob.cols, prob.rowptr)).tocsc(), prob.rhs)
    _, its_j, *_ = O.pcg_jacobi(prob.rowptr, prob.cols, prob.vals, prob.rhs, rtol=1e-10)
    assert reason == 2 and np.abs(u - x).max() <= 1e-8 * max(1.0, np.abs(x).max())
    assert its < 0.6 * its_j and its <= 30          # (one rank: 22; one hierarchy per rank would need 35-50)
