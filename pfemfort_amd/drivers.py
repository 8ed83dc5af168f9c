"""The reference's driver programs on top of the C ABI.

``tetrapoissonparallelimpl1`` / ``tetraelasticityparallelimpl1`` / ``triapoissonserialimpl1``
follow the PROGRAMs of the same name step by step (file:line cited inline): read or take a
mesh, Dirichlet bookkeeping, (re)numbering, ElemDofArray, pattern, element loop, solve,
gather.  Two element-loop modes:

* ``mode="batched"`` (default): the element loop is one device call (``pfem_assemble``).
* ``mode="compat"``: the loop is spelled out exactly like the Fortran -- one
  ``StiffnessResidual*`` call per element, ``MatSetValues`` / ``VecSetValues`` with
  ADD_VALUES, Dirichlet lifting on the host -- and only the solve runs on the GPU.  This
  is what an unchanged Fortran driver does through the Fortran shim.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from . import host as H
from .solver import ADD_VALUES, INSERT_VALUES, PetscSolver


@dataclass
class Result:
    kind: int
    mesh: H.Mesh
    dm: H.DofMap
    solver: PetscSolver
    soln_free: np.ndarray            # solution by free dof (NEW numbering), what temp.dat lists
    solnVTK: np.ndarray              # (nNode, ndof) by OLD node id, Dirichlet values filled in
    its: int
    reason: int
    rnorm: float
    timers: dict = field(default_factory=dict)

    def temp_dat(self):
        """Rows of the reference's ``temp.dat`` dump (tetrapoissonparallelimpl1.F:935-942):
        (ii, old node id, value), 1-based like the file (ndof=1) ."""
        assy = H.assy_for_soln(self.dm.NodeDofArrayNew)
        ndof = self.dm.NodeDofArrayNew.shape[1]
        if ndof == 1:
            ind = self.dm.node_map_get_old[assy] + 1
        else:   # tetraelasticityparallelimpl1.F:1034-1050
            ind = self.dm.node_map_get_old[assy // ndof] * ndof + assy % ndof + 1
        return np.arange(1, len(assy) + 1), ind, self.soln_free

    def write_outputs(self, directory=".", elem_procid=None):
        """The output step of the drivers: ``temp.dat`` (:935-942 / elasticity :1031-1046) and the
        legacy VTK file through the writervtk.F-compatible writer (:944-966)."""
        import os
        ndof = self.dm.NodeDofArrayNew.shape[1]
        ii, ind, val = self.temp_dat()
        H.write_temp_dat(os.path.join(directory, "temp.dat"), val, *((ii, ind) if ndof == 1 else (None, None)))
        name = "Poisson-soln.vtk" if ndof == 1 else "Elasticity-soln.vtk"
        pid = np.zeros(self.mesh.nElem, np.int32) if elem_procid is None else elem_procid
        H.writeoutputvtk(self.mesh.xyz.shape[0], self.mesh.xyz, self.mesh.conn, pid, self.solnVTK, os.path.join(directory, name),
                         ndof=ndof)
        return os.path.join(directory, name)


def _setup(kind, mesh: H.Mesh, nParts=1, node_proc_id=None):
    ndof = L.NDOF[kind]
    # NodeTypeOld / solnApplied / NodeDofArray*, node maps  (:316-367, :393-679)
    dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, nParts, node_proc_id)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)                         # :659-664, :832-838
    edof = H.elem_dof_array(conn_new, dm.NodeDofArrayNew)                # :698-713
    return dm, conn_new, xyz_new, edof


def _finish(kind, mesh, dm, solver, its, reason, rnorm, timers, u_free):
    ndof = L.NDOF[kind]
    full = dm.solnApplied.reshape(-1, ndof).copy()                       # :914-918 applied BC values
    assy = H.assy_for_soln(dm.NodeDofArrayNew)                           # :722-734
    full.reshape(-1)[assy] = u_free                                      # :936-941
    solnVTK = np.empty_like(full)
    solnVTK[dm.node_map_get_old] = full
    return Result(kind, mesh, dm, solver, u_free, solnVTK, its, reason, rnorm, timers)


def _run(kind, mesh, elemData, timeData, mode, rtol, maxits, verbose, pc=None):
    dm, conn_new, xyz_new, edof = _setup(kind, mesh)
    N = dm.size_global
    nsize = edof.shape[0]
    ndof = L.NDOF[kind]
    timers = {}
    solver = PetscSolver()
    n1 = min(50, N)                                                      # :762-773 (accepted, unused)
    solver.initialise(N, N, np.full(N, n1, np.int32), np.full(N, min(25, N), np.int32))   # :779
    solver.setTolerances(rtol=rtol, maxits=maxits)
    if pc:                                                               # KSPSetFromOptions: -pc_type (solverpetsc.F:191-206)
        solver.setPreconditioner(pc)
    if mode == "batched":
        solver.uploadMesh(kind, conn_new, xyz_new, edof, dm.solnApplied)
        solver.buildPattern()                                            # :786-802
        t0 = time.perf_counter()                                         # tstart :826
        solver.assemble(elemData, timeData)                              # setZero :817 + loop :828-884
        timers["assembly_s"] = time.perf_counter() - t0                  # :893
    elif mode == "compat":
        Kzero = np.zeros(nsize * nsize)
        for e in range(mesh.nElem):                                      # LoopElem :791-802
            f = edof[:, e]
            solver.MatSetValues(f, f, Kzero, INSERT_VALUES)
        solver.setZero()                                                 # :817
        valC = np.zeros(nsize)
        t0 = time.perf_counter()
        for e in range(mesh.nElem):                                      # :828-884
            nd = conn_new[:, e]
            xN, yN = xyz_new[0, nd], xyz_new[1, nd]
            if kind == L.POISSON_TET:
                K, F = H.StiffnessResidualPoissonLinearTetra(xN, yN, xyz_new[2, nd], elemData, timeData, valC)
            elif kind == L.ELAST_TET:
                K, F = H.StiffnessResidualElasticityLinearTetra(xN, yN, xyz_new[2, nd], elemData, timeData, valC)
            elif kind == L.POISSON_TRIA:
                K, F = H.StiffnessResidualPoissonLinearTria(xN, yN, elemData, timeData, valC)
            elif kind == L.ELAST_TRIA:
                K, F = H.StiffnessResidualElasticityLinearTria(xN, yN, elemData, timeData, valC)
            else:
                raise ValueError("compat mode needs a module element routine")
            f = edof[:, e]
            solver.MatSetValues(f, f, K.ravel(order="F"), ADD_VALUES)    # Fortran memory, read row-major
            for ii in range(nsize):                                      # LoopI/LoopJ :859-870
                if f[ii] == -1:
                    fact = dm.solnApplied[nd[ii // ndof] * ndof + ii % ndof]
                    for jj in range(nsize):
                        if f[jj] != -1:
                            F[jj] = F[jj] - K[jj, ii] * fact
            solver.VecSetValues(f, F, ADD_VALUES)                        # :880
        timers["assembly_s"] = time.perf_counter() - t0
    else:
        raise ValueError(mode)
    if mesh.force_node is not None and ndof > 1:                         # nodal forces :971-982, with the intended
        gdof = dm.NodeDofArrayNew[dm.node_map_get_new[mesh.force_node], mesh.force_dof]   # row = NodeDofArrayNew(n,d)-1
        if mode == "batched":
            solver.addNodalForces(gdof, mesh.force_val)
        else:
            for g, v in zip(gdof, mesh.force_val):
                solver.VecSetValues([g], [v], ADD_VALUES)
    t0 = time.perf_counter()                                             # :898
    its, reason, rnorm = solver.factoriseAndSolve()                      # :900
    timers["solve_s"] = time.perf_counter() - t0                         # :902
    if verbose:                                                          # solverpetsc.F:481-488
        print("Divergence." if reason < 0 else f" Convergence in {its} iterations.")
    u = solver.getSolution()                                             # VecScatterCreateToAll + VecGetArray :922-932
    timers.update(solver.timings())
    return _finish(kind, mesh, dm, solver, its, reason, rnorm, timers, u)


def run_parallel(kind, mesh: H.Mesh, elem_proc_id, node_proc_id, dist, torch, elemData=None, timeData=None, rtol=1e-5,
                 maxits=10000, staged=False, device=None, pc=None, spmv=None) -> Result:
    """The parallel branch of the driver PROGRAMs (tetrapoissonparallelimpl1.F:423-679 renumbering from
    (elem_proc_id, node_proc_id), :779 row blocks, :828-884 element loop over ``elem_proc_id == this_mpi_proc``,
    :898-902 solve, :922-932 VecScatterCreateToAll) on an initialised ``torch.distributed`` group, one rank per
    GPU.  ``staged``: ranks share a device and the group reduces host memory (gloo) -- tests only.
    Every rank returns the gathered Result (``soln_free`` is the whole solution, like vec_SEQ)."""
    from . import distributed as PD
    rank, world = dist.get_rank(), dist.get_world_size()
    ndof = L.NDOF[kind]
    if elemData is None:
        elemData = {L.ELAST_TET: H.ELAST_ELEMDATA, L.ELAST_TRIA: H.ELAST2D_ELEMDATA}.get(kind, H.POISSON_ELEMDATA)
    timeData = H.TIMEDATA if timeData is None else timeData
    dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, world, node_proc_id)
    conn_new, xyz_new = H.renumber_mesh(mesh, dm)                         # :659-664, :832-838
    mine = np.nonzero(np.asarray(elem_proc_id) == rank)[0]                # :829
    conn_loc = np.ascontiguousarray(conn_new[:, mine])
    edof = H.elem_dof_array(conn_loc, dm.NodeDofArrayNew)                # :698-713 for the rank's elements
    rs, re = int(dm.row_start[rank]), int(dm.row_end[rank])
    if device is None:
        device = torch.device("cuda", 0 if staged else rank % max(1, torch.cuda.device_count()))
    solver = PetscSolver().initialise(re - rs, dm.size_global, row_start=rs, device=device.index)   # :779
    solver.setTolerances(rtol=rtol, maxits=maxits)
    if pc:
        solver.setPreconditioner(pc)
    if spmv:
        solver.setSpmvFormat(spmv)
    timers = {}
    solver.uploadMesh(kind, conn_loc, xyz_new, edof, dm.solnApplied)
    hooks = PD.attach(solver, dist, torch, staged=staged)
    solver.buildPattern()                                                # :786-802
    t0 = time.perf_counter()
    solver.assemble(elemData, timeData)                                  # :817-884
    timers["assembly_s"] = time.perf_counter() - t0
    if mesh.force_node is not None and ndof > 1:                         # nodal forces :971-982 (intended row: see _run);
        gdof = dm.NodeDofArrayNew[dm.node_map_get_new[mesh.force_node], mesh.force_dof]   # every rank offers all of them,
        solver.addNodalForces(gdof, mesh.force_val)                      # the library keeps those of the rank's own rows
    t0 = time.perf_counter()
    its, reason, rnorm = solver.factoriseAndSolve()                      # :898-902
    timers["solve_s"] = time.perf_counter() - t0
    if hooks is not None and hooks.error is not None:
        raise hooks.error
    parts = [None] * world                                               # VecScatterCreateToAll (:922-932)
    dist.all_gather_object(parts, (rs, solver.getSolution()))
    u = np.empty(dm.size_global)
    for r0, x in parts:
        u[r0:r0 + len(x)] = x
    timers.update(solver.timings())
    return _finish(kind, mesh, dm, solver, its, reason, rnorm, timers, u)


def tetrapoissonparallelimpl1(mesh: H.Mesh | str, mode="batched", rtol=1e-5, maxits=10000, verbose=False, pc=None) -> Result:
    """PROGRAM TetraMeshPoissonEquation (tetrapoissonparallelimpl1.F) on one rank / one GPU."""
    if isinstance(mesh, str):
        mesh = H.read_mesh(mesh)
    return _run(L.POISSON_TET, mesh, H.POISSON_ELEMDATA, H.TIMEDATA, mode, rtol, maxits, verbose, pc)   # :822-824


def tetraelasticityparallelimpl1(mesh: H.Mesh | str, mode="batched", rtol=1e-5, maxits=10000, verbose=False, pc=None) -> Result:
    """PROGRAM of tetraelasticityparallelimpl1.F (body force only; DESIGN.md 'deviations')."""
    if isinstance(mesh, str):
        mesh = H.read_mesh(mesh)
    return _run(L.ELAST_TET, mesh, H.ELAST_ELEMDATA, H.TIMEDATA, mode, rtol, maxits, verbose, pc)       # :894-902


def triapoissonserialimpl1(mesh: H.Mesh | str, rtol=1e-5, maxits=10000, verbose=False, pc=None) -> Result:
    """PROGRAM of triapoissonserialimpl1.F: inline Ke = area*B*B^T (:573-594), Laplace."""
    if isinstance(mesh, str):
        mesh = H.read_mesh(mesh)
    return _run(L.POISSON_TRIA_INLINE, mesh, None, H.TIMEDATA, "batched", rtol, maxits, verbose, pc)


def triapoissonparallelimpl1(mesh: H.Mesh | str, mode="batched", rtol=1e-5, maxits=10000, verbose=False, pc=None) -> Result:
    """PROGRAM of triapoissonparallelimpl1.F on one rank: the module routine
    StiffnessResidualPoissonLinearTria (:862), kx = ky = 1, no source (next row 8f.1)."""
    if isinstance(mesh, str):
        mesh = H.read_mesh(mesh)
    return _run(L.POISSON_TRIA, mesh, np.array([1.0, 1.0]), H.TIMEDATA, mode, rtol, maxits, verbose, pc)


def triaelasticityparallelimpl1(mesh: H.Mesh | str, mode="batched", rtol=1e-5, maxits=10000, verbose=False, pc=None) -> Result:
    """PROGRAM of triaelasticityparallelimpl1.F on one rank with its intended semantics (the committed
    driver USEs a module that does not exist and reads thick/bforce uninitialised: SURVEY A.3#7, 8f.1):
    plane stress, E = 240.565, nu = 0.3 (REAL(4) literals), unit thickness, nodal forces from ForceBC."""
    if isinstance(mesh, str):
        mesh = H.read_mesh(mesh)
    return _run(L.ELAST_TRIA, mesh, H.ELAST2D_ELEMDATA, H.TIMEDATA, mode, rtol, maxits, verbose, pc)
