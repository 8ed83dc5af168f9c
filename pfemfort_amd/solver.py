"""``PetscSolver`` -- Python mirror of ``TYPE PetscSolver`` (MODULE Module_SolverPetsc,
solverpetsc.F:72-105) over the C ABI.  Same procedure names, argument meaning and status
machine as the reference; the linear algebra runs in hand-written HIP kernels on the GPU.
There is no CPU fallback: constructing the device object without a GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import (ADD_VALUES, ASSEMBLY_OK, FACTORISE_OK, INIT_OK, INSERT_VALUES,  # noqa: F401
                   PATTERN_OK, SOLVER_EMPTY)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class PetscSolver:
    """Drop-in for ``solverpetsc`` in the drivers (``call solverpetsc%initialise(...)`` ...)."""

    def __init__(self):
        self._h = C.c_void_p(None)
        self.nRow = self.nCol = 0
        self.its = 0
        self.reason = 0
        self.norm = 0.0
        self._keep = []          # ctypes callbacks / tensors that must outlive the handle

    # ---- solverpetsc.F:116-214 -------------------------------------------------------
    def initialise(self, size_local, size_global, diag_nnz=None, offdiag_nnz=None, row_start=0, device=-1):
        if self._h:
            self.free()
        dn = None if diag_nnz is None else _i32(diag_nnz)
        on = None if offdiag_nnz is None else _i32(offdiag_nnz)
        h = C.c_void_p(None)
        L.check(L.lib().pfem_solver_create(C.byref(h), size_local, size_global, row_start, _p(dn), _p(on), device),
                "PetscSolver%initialise")
        self._h = h
        self.nRow = self.nCol = size_global
        self.size_local, self.size_global, self.row_start = size_local, size_global, row_start
        return self

    @property
    def currentStatus(self):
        st = C.c_int(0)
        L.check(L.lib().pfem_solver_status(self._h, C.byref(st)), "PetscSolver%currentStatus")
        return st.value

    def setTolerances(self, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000):
        """KSPSetFromOptions stand-in (petsc_options.dat is not part of the reference tree)."""
        L.check(L.lib().pfem_solver_set_tolerances(self._h, rtol, abstol, dtol, maxits), "PetscSolver%setTolerances")

    def setStream(self, hip_stream):
        L.check(L.lib().pfem_solver_set_stream(self._h, C.c_void_p(hip_stream)), "PetscSolver%setStream")

    # ---- solverpetsc.F:222-246 -------------------------------------------------------
    def setZero(self):
        L.check(L.lib().pfem_solver_set_zero(self._h), "PetscSolver%setZero")

    # ---- solverpetsc.F:254-278 -------------------------------------------------------
    def free(self, collective=True):
        """``collective``: the explicit call of a multi-rank run, made by every rank while the process group lives -- the
        communication backend's teardown step (pfem_solver_comm_shutdown: a barrier of the peer-memory transport) runs first.
        The garbage collector's call is not collective."""
        if self._h:
            if collective and self._keep:            # (hooks are kept alive here: a backend was attached)
                L.lib().pfem_solver_comm_shutdown(self._h)
            L.lib().pfem_solver_destroy(self._h)
            self._h = C.c_void_p(None)
        self._keep.clear()

    def __del__(self):
        try:
            self.free(collective=False)
        except Exception:
            pass

    # ---- solverpetsc.F:286-320 -------------------------------------------------------
    def printInfo(self):
        L.check(L.lib().pfem_solver_print_info(self._h), "PetscSolver%printInfo")

    # ---- the PETSc calls the drivers make on solverpetsc%mtx / %rhsVec ---------------
    def MatSetValues(self, idxm, idxn, v, mode):
        """``MatSetValues(mtx,m,idxm,n,idxn,v,mode)``; ``v`` is read row-major like PETSc does
        (pass the Fortran-ordered Klocal's memory, i.e. ``np.asfortranarray(K).ravel(order='K')``)."""
        idxm = _i32(idxm); idxn = _i32(idxn)
        vv = None if v is None else _f64(v).ravel()
        L.check(L.lib().pfem_mat_set_values(self._h, len(idxm), _p(idxm), len(idxn), _p(idxn), _p(vv), mode),
                "MatSetValues")

    def VecSetValues(self, idx, v, mode):
        idx = _i32(idx)
        L.check(L.lib().pfem_vec_set_values(self._h, len(idx), _p(idx), _p(_f64(v)), mode), "VecSetValues")

    # ---- solverpetsc.F:328-401 -------------------------------------------------------
    def assembleMatrixAndVector(self, R, Cc, Klocal, Flocal):
        K = None if Klocal is None else np.asfortranarray(Klocal, dtype=np.float64)
        F = None if Flocal is None else _f64(Flocal)
        R = _i32(R); Cc = _i32(Cc)
        L.check(L.lib().pfem_solver_assemble_matrix_and_vector(self._h, len(R), _p(R), _p(Cc), _p(K), _p(F)),
                "PetscSolver%assembleMatrixAndVector")

    def assembleMatrix(self, R, Cc, Klocal):
        self.assembleMatrixAndVector(R, Cc, Klocal, None)

    def assembleVector(self, R, Flocal):
        self.assembleMatrixAndVector(R, R, None, Flocal)

    # ---- solverpetsc.F:409-509 -------------------------------------------------------
    def factorise(self):
        L.check(L.lib().pfem_solver_factorise(self._h), "PetscSolver%factorise")

    def _solve(self, fn, where):
        its = C.c_int(0); reason = C.c_int(0); rn = C.c_double(0)
        L.check(fn(self._h, C.byref(its), C.byref(reason), C.byref(rn)), where)
        self.its, self.reason, self.norm = its.value, reason.value, rn.value
        return self.its, self.reason, self.norm

    def solve(self):
        return self._solve(L.lib().pfem_solver_solve, "PetscSolver%solve")

    def factoriseAndSolve(self):
        return self._solve(L.lib().pfem_solver_factorise_and_solve, "PetscSolver%factoriseAndSolve")

    def getSolution(self):
        """VecGetArray(solnVec): the owned block of the solution."""
        x = np.empty(self.size_local)
        L.check(L.lib().pfem_solver_get_solution(self._h, _p(x)), "VecGetArray")
        return x

    def getHistory(self):
        h = np.empty(self.its + 1)
        n = C.c_int(0)
        L.check(L.lib().pfem_solver_get_history(self._h, _p(h), len(h), C.byref(n)), "pfem_solver_get_history")
        return h[:n.value]

    # ---- batched device path (the element loop as one call) --------------------------
    def uploadMesh(self, kind, conn, xyz, edof, solnApplied):
        conn = _i32(conn); xyz = _f64(xyz); edof = _i32(edof); sa = _f64(solnApplied)
        assert conn.shape[0] == L.NPELEM[kind] and xyz.shape[0] == L.NDIM[kind]
        assert edof.shape == (L.NPELEM[kind] * L.NDOF[kind], conn.shape[1])
        assert sa.size == xyz.shape[1] * L.NDOF[kind]
        L.check(L.lib().pfem_mesh_upload(self._h, kind, conn.shape[1], _p(conn), xyz.shape[1], _p(xyz), _p(edof), _p(sa)),
                "pfem_mesh_upload")
        self.kind = kind
        self.nElem = conn.shape[1]
        self.nNode = xyz.shape[1]

    def generateBoxMesh(self, kind, x0, x1, nEx, y0, y1, nEy, z0, z1, nEz, bc_mode=0, nparts=1, part=0, axis=2):
        """The structured box of genTetra.cpp + the driver's numbering for slab ``part`` of ``nparts`` along ``axis``
        (0 x, 1 y, 2 z, -1 the longest), generated on the device (the solver must have been initialised with
        ``host.box_slab_sizes`` for the same axis)."""
        L.check(L.lib().pfem_mesh_generate_box_axis(self._h, kind, x0, x1, nEx, y0, y1, nEy, z0, z1, nEz, bc_mode, axis, nparts, part),
                "pfem_mesh_generate_box_axis")
        self.kind = kind
        sz = L.lib().pfem_box_slab_sizes_axis
        n = C.c_int64(0); ne = C.c_int64(0)
        L.check(sz(nEx, nEy, nEz, bc_mode, L.NDOF[kind], axis, nparts, part, None, None, None, C.byref(n), C.byref(ne), None, None, None),
                "pfem_box_slab_sizes_axis")
        self.nElem, self.nNode = ne.value, n.value

    def downloadMesh(self):
        """(conn, xyz, edof_local, solnApplied) as the device holds them."""
        npe, ndof, ndim = L.NPELEM[self.kind], L.NDOF[self.kind], L.NDIM[self.kind]
        conn = np.empty((npe, self.nElem), np.int32); xyz = np.empty((ndim, self.nNode))
        edof = np.empty((npe * ndof, self.nElem), np.int32); sa = np.empty(self.nNode * ndof)
        L.check(L.lib().pfem_mesh_download(self._h, _p(conn), _p(xyz), _p(edof), _p(sa)), "pfem_mesh_download")
        return conn, xyz, edof, sa

    def buildPattern(self):
        L.check(L.lib().pfem_pattern_build(self._h), "pfem_pattern_build")

    def setAssemblyMode(self, mode):
        """"gather" (default: no atomics, bit-reproducible) or "scatter" (f64 atomics)."""
        L.check(L.lib().pfem_solver_set_assembly_mode(self._h, {"gather": 0, "scatter": 1}[mode]),
                "pfem_solver_set_assembly_mode")

    def assemblyInfo(self):
        """{"gather": bool, "hub_nodes": n, "block_threads": T, "lds_bytes": b} for the current pattern."""
        g = C.c_int(0); h = C.c_int(0); t = C.c_int(0); b = C.c_int64(0)
        L.check(L.lib().pfem_solver_assembly_info(self._h, C.byref(g), C.byref(h), C.byref(t), C.byref(b)), "pfem_solver_assembly_info")
        return {"gather": bool(g.value), "hub_nodes": h.value, "block_threads": t.value, "lds_bytes": b.value}

    def setSpmvFormat(self, fmt):
        """"auto" (16-bit column gaps when they fit; row groups when the pattern has them and the system is large),
        "grouped" (row groups at any size), "gaps16" (16-bit gaps, one row per lane) or "int32"."""
        L.check(L.lib().pfem_solver_set_spmv_format(self._h, {"auto": 0, "int32": 1, "gaps16": 2, "grouped": 3}[fmt]), "pfem_solver_set_spmv_format")

    def spmvColumnBits(self):
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_spmv_format(self._h, C.byref(b)), "pfem_solver_get_spmv_format")
        return b.value

    def spmvGapTable(self):
        """Entries of the table of distinct large gaps (dictionary form of the relative row groups), 0 otherwise."""
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_spmv_gap_table(self._h, C.byref(b)), "pfem_solver_get_spmv_gap_table")
        return b.value

    def spmvValueDictionary(self):
        """Distinct matrix values in the SpMV's dictionary when it streams 16-bit value codes (structured meshes), else 0."""
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_spmv_value_dictionary(self._h, C.byref(b)), "pfem_solver_get_spmv_value_dictionary")
        return b.value

    def amgValueDictionaries(self):
        """Per level of the last gamg hierarchy: distinct values in the level's dictionary when its SpMVs stream 16-bit value codes
        (0: fp64 values)."""
        n = C.c_int(0)
        e = (C.c_int * 32)()
        L.check(L.lib().pfem_solver_amg_value_dictionaries(self._h, 32, C.byref(n), e), "pfem_solver_amg_value_dictionaries")
        return list(e[:n.value])

    def spmvGapEscapes(self):
        """True when the row form streams 16-bit gaps with escapes to the int32 column array (k_spmv16e)."""
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_spmv_gap_escapes(self._h, C.byref(b)), "pfem_solver_get_spmv_gap_escapes")
        return bool(b.value)

    def setPreconditioner(self, pc):
        """"jacobi" (default; PCJACOBI), "pbjacobi" (node-block Jacobi, PETSc's -pc_type pbjacobi) or "gamg" (plain-aggregation
        multigrid V-cycle, PETSc's -pc_type gamg; one rank)."""
        L.check(L.lib().pfem_solver_set_preconditioner(self._h, {"jacobi": 0, "pbjacobi": 1, "gamg": 2}[pc]), "pfem_solver_set_preconditioner")

    def setSingleReduction(self, on=True):
        """KSPCGUseSingleReduction (-ksp_cg_single_reduction): one all-reduce per CG iteration instead of two
        (``None``: leave it to PFEM_CG_SINGLE_REDUCTION)."""
        L.check(L.lib().pfem_solver_set_cg_single_reduction(self._h, -1 if on is None else int(bool(on))),
                "pfem_solver_set_cg_single_reduction")

    def preconditioner(self):
        """The preconditioner the next solve uses ("pbjacobi" needs 3-dof row groups and one rank)."""
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_preconditioner(self._h, C.byref(b)), "pfem_solver_get_preconditioner")
        return ("jacobi", "pbjacobi", "gamg")[b.value]

    def setAmgOptions(self, cheb_degree=2, fine_degree=1, eig_ratio=None, coarse_scale=None):
        """-pc_gamg knobs: Chebyshev degree on the coarse levels / on the assembled matrix (0: the same), lmax/lmin of the
        smoothing interval, scaling of the coarse-grid correction (``None``: that knob stays automatic -- 16 / 1.5 for scalar
        problems and for displacement problems with rigid-body modes, 8 / 1.8 for 3-dof nodes without them: what
        amg_symbolic picks, pfem_amg.inc)."""
        L.check(L.lib().pfem_solver_set_amg_options(self._h, cheb_degree, fine_degree, -1.0 if eig_ratio is None else eig_ratio,
                                                    -1.0 if coarse_scale is None else coarse_scale), "pfem_solver_set_amg_options")

    def setAmgCycle(self, cycle=None):
        """-pc_mg_cycle_type: "v" (the default, also ``None``) or "w" (every coarse problem above the cycle's tail visited twice)."""
        code = {None: 0, "auto": 0, "v": 1, "w": 2}[cycle.lower() if isinstance(cycle, str) else cycle]
        L.check(L.lib().pfem_solver_set_amg_cycle(self._h, code), "pfem_solver_set_amg_cycle")

    def amgCycle(self):
        """What the last gamg solve ran: {"cycle": "v" | "w", "last_level_visited_twice": l} (0 with the V-cycle)."""
        c, w = C.c_int(0), C.c_int(0)
        L.check(L.lib().pfem_solver_amg_cycle(self._h, C.byref(c), C.byref(w)), "pfem_solver_amg_cycle")
        return {"cycle": "w" if c.value == 2 else "v", "last_level_visited_twice": w.value}

    def amgTransfer(self, level, xyz=False):
        """The transfer from ``level`` to the next: {"rbm": carries rotations?, "fine_bs", "coarse_bs", "dim", "n_nodes"} and, with
        ``xyz``, the level's node coordinates [3, n_nodes]."""
        rbm, fb, cb, dim = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        nn = C.c_int64(0)
        L.check(L.lib().pfem_solver_amg_transfer(self._h, level, C.byref(rbm), C.byref(fb), C.byref(cb), C.byref(dim), C.byref(nn), None), "pfem_solver_amg_transfer")
        out = {"rbm": bool(rbm.value), "fine_bs": fb.value, "coarse_bs": cb.value, "dim": dim.value, "n_nodes": nn.value}
        if xyz:
            a = np.empty((3, nn.value), np.float64)
            L.check(L.lib().pfem_solver_amg_transfer(self._h, level, C.byref(rbm), C.byref(fb), C.byref(cb), C.byref(dim), C.byref(nn), _p(a)), "pfem_solver_amg_transfer")
            out["xyz"] = a
        return out

    def incidencePatterns(self):
        """(patterns, longest list) of the gather form's incidence lists stored as translated copies; (0, 0): own records."""
        n = C.c_int(0); m = C.c_int(0)
        L.check(L.lib().pfem_solver_incidence_patterns(self._h, C.byref(n), C.byref(m)), "pfem_solver_incidence_patterns")
        return n.value, m.value

    def amgAggregation(self):
        """How every level's aggregates were formed: a list of "bricks" / "node-bricks" / "split-bricks" / "lattice-passes" /
        "matching" / "roots" per transfer (one fewer than levels)."""
        mx = 32
        nl = C.c_int(0)
        kind = (C.c_int * mx)()
        L.check(L.lib().pfem_solver_amg_aggregation(self._h, mx, C.byref(nl), kind), "pfem_solver_amg_aggregation")
        names = {0: "none", 1: "bricks", 2: "node-bricks", 3: "split-bricks", 4: "lattice-passes", 5: "matching", 6: "roots"}
        return [names.get(kind[l], str(kind[l])) for l in range(max(nl.value - 1, 0))]

    def amgInfo(self):
        """The multigrid hierarchy of the last ``gamg`` solve: rows / nonzeros / eigenvalue bound per level, phase times."""
        mx = 32
        nl = C.c_int(0); rows = (C.c_int64 * mx)(); nnz = (C.c_int64 * mx)(); lam = (C.c_double * mx)()
        sym = C.c_double(0); num = C.c_double(0); deg = C.c_int(0); fdeg = C.c_int(0); ratio = C.c_double(0); scale = C.c_double(0)
        L.check(L.lib().pfem_solver_amg_info(self._h, mx, C.byref(nl), rows, nnz, lam, C.byref(sym), C.byref(num), C.byref(deg),
                                             C.byref(fdeg), C.byref(ratio), C.byref(scale)), "pfem_solver_amg_info")
        n = nl.value
        return {"levels": n, "rows": list(rows[:n]), "nnz": list(nnz[:n]), "lambda_max": list(lam[:n]), "symbolic_ms": sym.value,
                "numeric_ms": num.value, "cheb_degree": deg.value, "fine_degree": fdeg.value, "eig_ratio": ratio.value, "coarse_scale": scale.value}

    def amgAggregates(self, level, n_rows):
        """Coarse dof of every dof of ``level`` (``n_rows`` = amgInfo()["rows"][level])."""
        a = np.empty(n_rows, np.int32)
        L.check(L.lib().pfem_solver_amg_aggregates(self._h, level, _p(a)), "pfem_solver_amg_aggregates")
        return a

    def amgLayout(self):
        """Several ranks: {"coupled": one hierarchy across the ranks?, "distributed_levels": how many of its levels are spread
        over the ranks (the rest is held whole by every rank), "first_dof": [...], "local_rows": [...]} per level."""
        mx = 32
        cp, nd = C.c_int(0), C.c_int(0)
        first = np.zeros(mx, np.int64)
        loc = np.zeros(mx, np.int64)
        L.check(L.lib().pfem_solver_amg_layout(self._h, mx, C.byref(cp), C.byref(nd), _p(first), _p(loc)), "pfem_solver_amg_layout")
        nl = self.amgInfo()["levels"]
        ex, ar = C.c_int(0), C.c_int(0)
        L.check(L.lib().pfem_solver_amg_comm_counts(self._h, C.byref(ex), C.byref(ar)), "pfem_solver_amg_comm_counts")
        lat = C.c_int(0)
        L.check(L.lib().pfem_solver_amg_pairing(self._h, C.byref(lat)), "pfem_solver_amg_pairing")
        return {"coupled": bool(cp.value), "distributed_levels": nd.value, "first_dof": first[:nl].tolist(), "local_rows": loc[:nl].tolist(),
                "exchanges_per_cycle": ex.value, "allreduces_per_cycle": ar.value, "lattice_levels": lat.value}

    def amgCycleProfile(self):
        """One instrumented V-cycle of the hierarchy across the ranks (collective): per level the exchanges, their time and bytes;
        the cycle's all-reduces; the cycle's time."""
        mx = 32
        nl, na = C.c_int(0), C.c_int(0)
        ex = np.zeros(mx, np.int32); ms = np.zeros(mx, np.float64); dbl = np.zeros(mx, np.int64)
        ams, adbl, cyc = C.c_double(0), C.c_int64(0), C.c_double(0)
        L.check(L.lib().pfem_solver_amg_cycle_profile(self._h, mx, C.byref(nl), _p(ex), _p(ms), _p(dbl), C.byref(na), C.byref(ams), C.byref(adbl),
                                                      C.byref(cyc)), "pfem_solver_amg_cycle_profile")
        n = nl.value
        return {"cycle_ms": cyc.value, "exchanges_per_level": ex[:n].tolist(), "exchange_ms_per_level": ms[:n].tolist(),
                "exchange_bytes_per_level": (8 * dbl[:n]).tolist(), "allreduces": na.value, "allreduce_ms": ams.value, "allreduce_bytes": 8 * adbl.value}

    def spmvRowGroup(self):
        """Rows served by one lane of the current SpMV (3: row-grouped form)."""
        b = C.c_int(0)
        L.check(L.lib().pfem_solver_get_spmv_row_group(self._h, C.byref(b)), "pfem_solver_get_spmv_row_group")
        return b.value

    def spmvFormatBytes(self):
        """Bytes one launch of the selected SpMV form moves at best (its storage + x + y)."""
        b = C.c_int64(0)
        L.check(L.lib().pfem_solver_spmv_bytes(self._h, C.byref(b)), "pfem_solver_spmv_bytes")
        return b.value

    def assemble(self, elemData, timeData):
        ed = None if elemData is None else _f64(elemData)
        L.check(L.lib().pfem_assemble(self._h, _p(ed), _p(_f64(timeData))), "pfem_assemble")

    def addNodalForces(self, global_dof, values):
        """VecSetValue(rhsVec, row, fact, ADD_VALUES) loop of the elasticity drivers (:971-982)."""
        g = np.ascontiguousarray(global_dof, dtype=np.int64); v = _f64(values)
        L.check(L.lib().pfem_rhs_add_values(self._h, len(g), _p(g), _p(v)), "pfem_rhs_add_values")

    def evalElems(self, elemData, timeData):
        ns = L.NPELEM[self.kind] * L.NDOF[self.kind]
        K = np.empty((self.nElem, ns, ns)); F = np.empty((self.nElem, ns))
        ed = None if elemData is None else _f64(elemData)
        L.check(L.lib().pfem_eval_elems(self._h, _p(ed), _p(_f64(timeData)), _p(K), _p(F)), "pfem_eval_elems")
        return K.transpose(0, 2, 1).copy(), F     # column-major blocks -> K[e][i,j] = Klocal(i,j)

    def matrixInfo(self):
        a, b, c, d = (C.c_int64(0) for _ in range(4))
        L.check(L.lib().pfem_matrix_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "pfem_matrix_info")
        return {"n_owned": a.value, "n_local": b.value, "nnz": c.value, "stored": d.value}

    def localToGlobal(self):
        g = np.empty(self.matrixInfo()["n_local"], np.int64)
        L.check(L.lib().pfem_get_local_to_global(self._h, _p(g)), "pfem_get_local_to_global")
        return g

    def getCSR(self, values=True):
        info = self.matrixInfo()
        rowptr = np.empty(info["n_local"] + 1, np.int64)
        cols = np.empty(info["nnz"], np.int32)
        vals = np.empty(info["nnz"]) if values else None
        L.check(L.lib().pfem_get_csr(self._h, _p(rowptr), _p(cols), _p(vals)), "pfem_get_csr")
        return rowptr, cols, vals

    def getRHS(self):
        r = np.empty(self.matrixInfo()["n_local"])
        L.check(L.lib().pfem_get_rhs(self._h, _p(r)), "pfem_get_rhs")
        return r

    def spmv(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        L.check(L.lib().pfem_spmv(self._h, _p(x), _p(y)), "pfem_spmv")
        return y

    def benchSpmv(self, reps=20):
        ms = C.c_double(0)
        L.check(L.lib().pfem_bench_spmv(self._h, reps, C.byref(ms)), "pfem_bench_spmv")
        return ms.value

    def profileSpmv(self, enable=True):
        L.check(L.lib().pfem_solver_profile_spmv(self._h, int(enable)), "pfem_solver_profile_spmv")

    def timings(self):
        t = L.Timings()
        L.check(L.lib().pfem_get_timings(self._h, C.byref(t)), "pfem_get_timings")
        return {k: getattr(t, k) for k, _ in t._fields_}

    # ---- multi-GPU plumbing ----------------------------------------------------------
    def ghosts(self):
        n = C.c_int64(0)
        L.check(L.lib().pfem_get_ghosts(self._h, C.byref(n), None), "pfem_get_ghosts")
        g = np.empty(n.value, np.int64)
        L.check(L.lib().pfem_get_ghosts(self._h, C.byref(n), _p(g)), "pfem_get_ghosts")
        return g

    def setNeighbours(self, peers, peer_off, shared_gid):
        """Install the neighbour plan (``host.neighbour_plan``)."""
        pe = _i32(peers); off = np.ascontiguousarray(peer_off, dtype=np.int64); g = np.ascontiguousarray(shared_gid, dtype=np.int64)
        L.check(L.lib().pfem_solver_set_neighbours(self._h, len(pe), _p(pe), _p(off), _p(g)), "pfem_solver_set_neighbours")

    def setCommRccl(self, rank, nranks, unique_id: bytes):
        """RCCL bound inside the library; ``unique_id`` = the bytes of ``rccl_unique_id()`` from rank 0."""
        assert len(unique_id) == L.RCCL_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), L.RCCL_ID_BYTES)
        L.check(L.lib().pfem_solver_set_comm_rccl(self._h, rank, nranks, buf), "pfem_solver_set_comm_rccl")

    def setCommHost(self, rank, nranks, allreduce_cb, exchange_cb):
        """Host-memory hooks (gloo, MPI): ``allreduce_cb(ctx, buf, count)``, ``exchange_cb(ctx, n, peers, off, send, recv)``."""
        a = L.HOST_ALLREDUCE_FN(allreduce_cb) if allreduce_cb is not None else L.HOST_ALLREDUCE_FN()
        e = L.HOST_EXCHANGE_FN(exchange_cb) if exchange_cb is not None else L.HOST_EXCHANGE_FN()
        self._keep.extend([a, e])
        L.check(L.lib().pfem_solver_set_comm_host(self._h, rank, nranks, a, e, None), "pfem_solver_set_comm_host")

    def setCommPeer(self, rank, nranks, allreduce_cb, exchange_cb=None):
        """Peer-memory transport (ranks map each other's receive boxes through hipIpc*; kernels write into them directly): the
        device-side path between processes that share a GPU.  The host hooks carry the bring-up and oversized all-reduces."""
        a = L.HOST_ALLREDUCE_FN(allreduce_cb) if allreduce_cb is not None else L.HOST_ALLREDUCE_FN()
        e = L.HOST_EXCHANGE_FN(exchange_cb) if exchange_cb is not None else L.HOST_EXCHANGE_FN()
        self._keep.extend([a, e])
        L.check(L.lib().pfem_solver_set_comm_peer(self._h, rank, nranks, a, e, None), "pfem_solver_set_comm_peer")

    def commSelftest(self, count=1000):
        """Collective transport check (stamped exchange with every other rank + a known all-reduce): wrong entries."""
        bad = C.c_int64(0)
        L.check(L.lib().pfem_solver_comm_selftest(self._h, count, C.byref(bad)), "pfem_solver_comm_selftest")
        return bad.value

    def commBench(self, count, reps=200, allreduce_count=4, slab_neighbours=False):
        """Collective transport timing: (ms per exchange of ``count`` doubles with every other rank -- ``slab_neighbours``: with rank - 1
        and rank + 1 only --, ms per all-reduce of ``allreduce_count`` doubles)."""
        a, b = C.c_double(0), C.c_double(0)
        L.check(L.lib().pfem_solver_comm_bench_sizes(self._h, count, allreduce_count, reps, 1 if slab_neighbours else 0, C.byref(a), C.byref(b)),
                "pfem_solver_comm_bench_sizes")
        return a.value, b.value

    def commInfo(self):
        npe = C.c_int(0); d = C.c_int64(0); b = C.c_int64(0); t = C.c_int64(0)
        L.check(L.lib().pfem_solver_comm_info(self._h, C.byref(npe), C.byref(d), C.byref(b), C.byref(t)), "pfem_solver_comm_info")
        return {"n_peers": npe.value, "doubles_per_exchange": d.value, "boundary_slices": b.value, "total_slices": t.value}


    def commDescribe(self):
        """What carries the multi-rank solve, as the transport reports it: backend ("rccl" / "host" / "none"), the rank
        count / device / version of the bound RCCL communicators (ncclCommCount, ncclCommCuDevice, ncclGetVersion; -1
        for host hooks), the solver's device, the agreed SpMV form (0 in order, 1 overlapped, -1 not voted yet)."""
        name = C.create_string_buffer(32)
        r, d, v, sd, ov = (C.c_int(0) for _ in range(5))
        L.check(L.lib().pfem_solver_comm_describe(self._h, name, 32, C.byref(r), C.byref(d), C.byref(v), C.byref(sd), C.byref(ov)),
                "pfem_solver_comm_describe")
        return {"backend": name.value.decode(), "backend_ranks": r.value, "backend_device": d.value, "backend_version": v.value,
                "solver_device": sd.value, "overlapped_form": ov.value}


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and broadcasts the bytes)."""
    buf = C.create_string_buffer(L.RCCL_ID_BYTES)
    L.check(L.lib().pfem_rccl_unique_id(buf), "pfem_rccl_unique_id")
    return buf.raw
