"""ctypes/numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module (it is the checker, never the thing measured or shipped).

* ``liboracle.so``        -- oracle/pfem_oracle.c, the C restatement (always available)
* ``_ref/libpfem_ref.so`` -- the reference's own Fortran element routines compiled in place
  with flang (optional: built only where /root/reference exists; see oracle/Makefile)

Array conventions follow the Fortran drivers (column-major == SoA):
``xyz`` is ``(ndim, nNode)`` C-contiguous, ``conn`` / ``edof`` are ``(npElem|nsize, nElem)``
C-contiguous int32, 0-based, ``-1`` marks a Dirichlet dof.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POISSON_TRIA, POISSON_TET, ELAST_TET, POISSON_TRIA_INLINE, ELAST_TRIA = 1, 2, 3, 4, 5
NPELEM = {POISSON_TRIA: 3, POISSON_TET: 4, ELAST_TET: 4, POISSON_TRIA_INLINE: 3, ELAST_TRIA: 3}
NDOF = {POISSON_TRIA: 1, POISSON_TET: 1, ELAST_TET: 3, POISSON_TRIA_INLINE: 1, ELAST_TRIA: 2}
NDIM = {POISSON_TRIA: 2, POISSON_TET: 3, ELAST_TET: 3, POISSON_TRIA_INLINE: 2, ELAST_TRIA: 2}

# REAL(4) literals of the drivers widened to double
F32 = lambda v: float(np.float32(v))  # noqa: E731
POISSON_ELEMDATA = np.array([1.0, 1.0, 1.0])                     # tetrapoissonparallelimpl1.F:822
ELAST_ELEMDATA = np.array([F32(240.565), F32(0.3), 1.0, F32(0.1), 0.0, 0.0])  # tetraelasticity...F:895-899
TIMEDATA = np.array([0.0, 1.0, 0.0])                             # :823
# triaelasticityparallelimpl1.F:907 sets only E, nu (thick, bforce are left uninitialised there:
# SURVEY 8f.1); intended values: unit thickness, no body force
ELAST2D_ELEMDATA = np.array([F32(240.565), F32(0.3), 1.0, 0.0, 0.0, 0.0])


def build(ref: bool = True) -> None:
    """Compile liboracle.so (and _ref when the reference tree + flang exist)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
    return _lib


def ref_available() -> bool:
    """Is the flang-compiled reference built?  Does NOT load it (a skipif may ask at import time)."""
    return os.path.exists(os.path.join(_HERE, "_ref", "libpfem_ref.so"))


def ref_lib():
    """The flang-compiled reference routines, or None when not built.  Loaded on first use only, by the CPU tests that
    pin the restatement against it; nothing in the GPU suite calls this."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libpfem_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ----------------------------------------------------------------------------
# mesh files and structured generator
# ----------------------------------------------------------------------------
def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")


@dataclass
class Mesh:
    xyz: np.ndarray        # (ndim, nNode) float64
    conn: np.ndarray       # (npElem, nElem) int32, 0-based OLD numbering
    bc_node: np.ndarray    # (nDBC,) int32 0-based
    bc_dof: np.ndarray     # (nDBC,) int32 0-based
    bc_val: np.ndarray     # (nDBC,) float64

    @property
    def nNode(self):
        return self.xyz.shape[1]

    @property
    def nElem(self):
        return self.conn.shape[1]


def read_mesh(prefix: str) -> Mesh:
    """Read ``<prefix>-nodes.dat[.gz]``, ``-elems``, ``-DirichBC`` (1-based ASCII)."""
    def find(kind):
        for ext in (".dat.gz", ".dat"):
            if os.path.exists(prefix + "-" + kind + ext):
                return prefix + "-" + kind + ext
        raise FileNotFoundError(prefix + "-" + kind)
    with _open(find("nodes")) as f:
        nodes = np.loadtxt(f, ndmin=2)
    with _open(find("elems")) as f:
        elems = np.loadtxt(f, dtype=np.int64, ndmin=2)
    with _open(find("DirichBC")) as f:
        bcs = np.loadtxt(f, ndmin=2)
    xyz = np.ascontiguousarray(nodes[:, 1:].T)
    conn = np.ascontiguousarray((elems[:, 1:] - 1).T.astype(np.int32))
    return Mesh(xyz, conn, (bcs[:, 0] - 1).astype(np.int32), (bcs[:, 1] - 1).astype(np.int32),
                bcs[:, 2].copy())


def gen_box_tets(x0, x1, nEx, y0, y1, nEy, z0, z1, nEz, bc_mode=0, ndof=1) -> Mesh:
    """genTetra.cpp restated (oracle/pfem_oracle.c: orc_gen_box_tets)."""
    nNode = (nEx + 1) * (nEy + 1) * (nEz + 1)
    nElem = 6 * nEx * nEy * nEz
    xyz = np.empty((3, nNode))
    conn = np.empty((4, nElem), dtype=np.int32)
    n = C.c_int64(0)
    args = (C.c_double(x0), C.c_double(x1), C.c_int(nEx), C.c_double(y0), C.c_double(y1), C.c_int(nEy),
            C.c_double(z0), C.c_double(z1), C.c_int(nEz), C.c_int(bc_mode), C.c_int(ndof))
    rc = lib().orc_gen_box_tets(*args, None, None, C.byref(n), None, None, None)
    assert rc == 0
    bn = np.empty(n.value, np.int32); bd = np.empty(n.value, np.int32); bv = np.empty(n.value)
    rc = lib().orc_gen_box_tets(*args, _p(xyz), _p(conn), C.byref(n), _p(bn), _p(bd), _p(bv))
    assert rc == 0
    return Mesh(xyz, conn, bn, bd, bv)


# ----------------------------------------------------------------------------
# element routines
# ----------------------------------------------------------------------------
def eval_elems(kind, xyz, conn, elemData, timeData=TIMEDATA):
    """Ke (nElem, nsize, nsize) [K[e].T is the Fortran Klocal -> we return K[e][i,j]=Klocal(i,j)],
    Fe (nElem, nsize)."""
    xyz = _f64(xyz); conn = _i32(conn)
    nElem = conn.shape[1]; nNode = xyz.shape[1]
    ns = NPELEM[kind] * NDOF[kind]
    K = np.empty((nElem, ns, ns)); F = np.empty((nElem, ns))
    ed = _f64(np.resize(elemData, 6) if len(elemData) < 6 else elemData)
    rc = lib().orc_eval_elems(C.c_int(kind), C.c_int64(nElem), _p(conn), C.c_int64(nNode), _p(xyz),
                              _p(ed), _p(_f64(timeData)), _p(K), _p(F))
    if rc:
        raise RuntimeError(f"oracle element evaluation failed rc={rc}")
    return K.transpose(0, 2, 1).copy(), F          # column-major blocks -> [i,j]


def ref_eval_elems(kind, xyz, conn, elemData, timeData=TIMEDATA):
    """Same through the flang-compiled reference routines (oracle/_ref)."""
    L = ref_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libpfem_ref.so not built")
    xyz = _f64(xyz); conn = _i32(conn)
    nElem = conn.shape[1]
    ns = NPELEM[kind] * NDOF[kind]
    g = [np.ascontiguousarray(xyz[d][conn].T) for d in range(NDIM[kind])]   # (nElem, npElem)
    K = np.empty((nElem, ns, ns)); F = np.empty((nElem, ns))
    ed = _f64(elemData); td = _f64(timeData)
    if kind == POISSON_TET:
        L.ref_poisson_tet_batch(C.c_int64(nElem), _p(g[0]), _p(g[1]), _p(g[2]), _p(ed), _p(td), _p(K), _p(F))
    elif kind == ELAST_TET:
        L.ref_elast_tet_batch(C.c_int64(nElem), _p(g[0]), _p(g[1]), _p(g[2]), _p(ed), _p(td), _p(K), _p(F))
    elif kind == POISSON_TRIA:
        L.ref_poisson_tria_batch(C.c_int64(nElem), _p(g[0]), _p(g[1]), _p(ed), _p(td), _p(K), _p(F))
    elif kind == ELAST_TRIA:
        L.ref_elast_tria_batch(C.c_int64(nElem), _p(g[0]), _p(g[1]), _p(ed), _p(td), _p(K), _p(F))
    else:
        raise ValueError(kind)
    return K.transpose(0, 2, 1).copy(), F


# ----------------------------------------------------------------------------
# bookkeeping
# ----------------------------------------------------------------------------
@dataclass
class DofMap:
    node_map_get_old: np.ndarray
    node_map_get_new: np.ndarray
    NodeDofArrayNew: np.ndarray     # (nNode, ndof) 0-based, -1 = Dirichlet
    solnApplied: np.ndarray         # (nNode*ndof,) indexed by NEW node*ndof+d
    node_start: np.ndarray
    node_end: np.ndarray
    row_start: np.ndarray
    row_end: np.ndarray
    size_global: int


def dof_numbering(nNode, ndof, bc_node, bc_dof, bc_val, nParts=1, node_proc_id=None) -> DofMap:
    old = np.empty(nNode, np.int32); new = np.empty(nNode, np.int32)
    nda = np.empty((nNode, ndof), np.int32); sa = np.empty(nNode * ndof)
    ns = np.zeros(nParts, np.int64); ne = np.zeros(nParts, np.int64)
    rs = np.zeros(nParts, np.int64); re = np.zeros(nParts, np.int64)
    sg = C.c_int64(0)
    npid = None if node_proc_id is None else _i32(node_proc_id)
    rc = lib().orc_dof_numbering(C.c_int64(nNode), C.c_int(ndof), C.c_int64(len(bc_node)), _p(_i32(bc_node)),
                                 _p(_i32(bc_dof)), _p(_f64(bc_val)), C.c_int(nParts), _p(npid), _p(old), _p(new),
                                 _p(nda), _p(sa), _p(ns), _p(ne), _p(rs), _p(re), C.byref(sg))
    assert rc == 0, rc
    return DofMap(old, new, nda, sa, ns, ne, rs, re, sg.value)


def elem_dof_array(conn_new, NodeDofArrayNew):
    conn_new = _i32(conn_new)
    npE, nElem = conn_new.shape
    ndof = NodeDofArrayNew.shape[1]
    edof = np.empty((npE * ndof, nElem), np.int32)
    lib().orc_elem_dof_array(C.c_int64(nElem), C.c_int(npE), C.c_int(ndof), _p(conn_new),
                             _p(_i32(NodeDofArrayNew)), _p(edof))
    return edof


def assy_for_soln(NodeDofArrayNew):
    nNode, ndof = NodeDofArrayNew.shape
    n = int((NodeDofArrayNew >= 0).sum())
    out = np.empty(n, np.int32)
    lib().orc_assy_for_soln(C.c_int64(nNode), C.c_int(ndof), _p(_i32(NodeDofArrayNew)), _p(out))
    return out


# ----------------------------------------------------------------------------
# pattern, assembly, solve
# ----------------------------------------------------------------------------
def csr_pattern(edof, N):
    edof = _i32(edof)
    nsize, nElem = edof.shape
    rowptr = np.empty(N + 1, np.int64)
    rc = lib().orc_csr_pattern(C.c_int64(nElem), C.c_int(nsize), _p(edof), C.c_int64(N), _p(rowptr), None)
    assert rc == 0
    cols = np.empty(rowptr[-1], np.int32)
    rc = lib().orc_csr_pattern(C.c_int64(nElem), C.c_int(nsize), _p(edof), C.c_int64(N), _p(rowptr), _p(cols))
    assert rc == 0
    return rowptr, cols


def assemble(kind, xyz_new, conn_new, edof, solnApplied, elemData, N, rowptr, cols,
             timeData=TIMEDATA, elem_proc_id=None, part=0):
    xyz_new = _f64(xyz_new); conn_new = _i32(conn_new); edof = _i32(edof)
    vals = np.zeros(len(cols)); rhs = np.zeros(N)
    ed = _f64(np.resize(elemData, 6) if len(elemData) < 6 else elemData)
    epid = None if elem_proc_id is None else _i32(elem_proc_id)
    rc = lib().orc_assemble(C.c_int(kind), C.c_int64(conn_new.shape[1]), _p(conn_new), C.c_int64(xyz_new.shape[1]),
                            _p(xyz_new), _p(edof), _p(_f64(solnApplied)), _p(ed), _p(_f64(timeData)),
                            _p(epid), C.c_int(part), C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(rhs))
    if rc:
        raise RuntimeError(f"oracle assembly failed rc={rc}")
    return vals, rhs


def assemble_mt(kind, xyz_new, conn_new, edof, solnApplied, elemData, N, rowptr, cols, timeData=TIMEDATA):
    """OpenMP + atomics variant, for bench.py's cpu_baseline timing only (sum order not fixed)."""
    xyz_new = _f64(xyz_new); conn_new = _i32(conn_new); edof = _i32(edof)
    vals = np.zeros(len(cols)); rhs = np.zeros(N)
    ed = _f64(np.resize(elemData, 6) if len(elemData) < 6 else elemData)
    rc = lib().orc_assemble_mt(C.c_int(kind), C.c_int64(conn_new.shape[1]), _p(conn_new), C.c_int64(xyz_new.shape[1]),
                               _p(xyz_new), _p(edof), _p(_f64(solnApplied)), _p(ed), _p(_f64(timeData)),
                               C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(rhs))
    if rc:
        raise RuntimeError(f"oracle assembly failed rc={rc}")
    return vals, rhs


def set_threads(n):
    """OpenMP thread count of liboracle.so (its own libgomp instance)."""
    import ctypes.util
    g = C.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1")
    g.omp_set_num_threads(int(n))


def stream_triad_gbps(n=1 << 27, reps=5):
    """Host memory bandwidth (STREAM triad, GB/s) with the current OpenMP thread count: what the CPU baseline is bound by."""
    f = lib().orc_stream_triad_gbps
    f.restype = C.c_double
    return float(f(C.c_int64(n), C.c_int(reps)))


def spmv(rowptr, cols, vals, x):
    N = len(rowptr) - 1
    y = np.empty(N)
    lib().orc_spmv(C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(_f64(x)), _p(y))
    return y


def pcg_jacobi(rowptr, cols, vals, b, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000, hist_len=0):
    N = len(rowptr) - 1
    x = np.empty(N)
    its = C.c_int(0); reason = C.c_int(0); rn = C.c_double(0)
    hist = np.zeros(max(hist_len, 1))
    rc = lib().orc_pcg_jacobi(C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(_f64(b)), _p(x),
                              C.c_double(rtol), C.c_double(abstol), C.c_double(dtol), C.c_int(maxits),
                              C.byref(its), C.byref(reason), C.byref(rn), _p(hist), C.c_int(hist_len))
    assert rc == 0
    return x, its.value, reason.value, rn.value, hist[:min(hist_len, its.value + 1)]


def pcg_jacobi_single_reduction(rowptr, cols, vals, b, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000, hist_len=0):
    """Jacobi-PCG in PETSc's single-reduction form (KSPCGUseSingleReduction)."""
    N = len(rowptr) - 1
    x = np.empty(N)
    its = C.c_int(0); reason = C.c_int(0); rn = C.c_double(0)
    hist = np.zeros(max(hist_len, 1))
    rc = lib().orc_pcg_jacobi_single_reduction(C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(_f64(b)), _p(x),
                                               C.c_double(rtol), C.c_double(abstol), C.c_double(dtol), C.c_int(maxits),
                                               C.byref(its), C.byref(reason), C.byref(rn), _p(hist), C.c_int(hist_len))
    assert rc == 0
    return x, its.value, reason.value, rn.value, hist[:min(hist_len, its.value + 1)]


def pcg_bjacobi_ilu0(rowptr, cols, vals, b, block_start=None, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000):
    """CG with PETSc's PCBJACOBI default (ILU(0) on each rank's diagonal block, natural ordering): what the
    reference's solverpetsc.F:187,206 sets.  ``block_start`` = row-block boundaries (default: one block)."""
    N = len(rowptr) - 1
    bs = np.ascontiguousarray([0, N] if block_start is None else block_start, dtype=np.int64)
    x = np.empty(N)
    its = C.c_int(0); reason = C.c_int(0); rn = C.c_double(0)
    rc = lib().orc_pcg_bjacobi_ilu0(C.c_int64(N), _p(rowptr), _p(cols), _p(vals), _p(_f64(b)), _p(x), C.c_int(len(bs) - 1),
                                    _p(bs), C.c_double(rtol), C.c_double(abstol), C.c_double(dtol), C.c_int(maxits),
                                    C.byref(its), C.byref(reason), C.byref(rn))
    assert rc == 0
    return x, its.value, reason.value, rn.value


def _lat_sim_passes(hi, axis, passes):
    """The axis sequence of one level's passes: x, y, z, x, ... skipping an axis without extent (pfem_amg.inc: lat_sim_passes)."""
    hi, shift = list(hi), [0, 0, 0]
    for _ in range(passes):
        k = 0
        while k < 3 and hi[axis] < 1:
            axis = (axis + 1) % 3
            k += 1
        if k == 3:
            break
        shift[axis] += 1
        hi[axis] >>= 1
        axis = (axis + 1) % 3
    return shift, hi, axis


def _lat_pad(cuts, single, hi, occ, axis, passes):
    """Positions 0..hi[d] in stretches of one owner starting at cuts[d][k] -> padded positions in which no brick of the level holds
    positions of two stretches (pfem_amg.inc: lat_pad).  The odd position a stretch of odd length leaves over goes to the
    stretch's START (which then begins on an odd position, behind a free one) when the last odd one went to the end, and the
    other way round (``single[d][k]`` = 1: the last one went to the start); the first stretch always starts at 0, axes halved
    twice in a level keep the plain alignment.  (tables, new cuts, new single, new hi) or None."""
    shift = [0, 0, 0]
    for _ in range(5):
        table, ncuts, nsingle, nhi = [np.zeros(1024, np.int64) for _ in range(3)], [[], [], []], [[], [], []], [0, 0, 0]
        for d in range(3):
            nxt, cs = 0, (list(cuts[d]) or [0])
            for k, start in enumerate(cs):
                end = cs[k + 1] if k + 1 < len(cs) else hi[d] + 1
                ln = end - start
                ln_occ = min(end, occ[1][d] + 1) - max(start, occ[0][d])          # occupied positions: a Dirichlet plane has a position and no dofs
                prev = single[d][k] if k < len(single[d]) else 0
                a = 1 << shift[d]
                base = 0 if k == 0 else (nxt + a - 1) // a * a
                now = prev if shift[d] == 0 else 0
                if k > 0 and shift[d] == 1 and ln_occ > 0 and ln_occ % 2 == 1 and prev == 0:
                    base, now = (nxt + 1) | 1, 1
                ncuts[d].append(base)
                nsingle[d].append(now)
                if base + ln > 1024:
                    return None
                table[d][start:end] = base + np.arange(ln)
                nxt = base + ln
            nhi[d] = nxt - 1
        s2, _, _ = _lat_sim_passes(nhi, axis, passes)
        if all(s2[d] <= shift[d] for d in range(3)):
            return table, ncuts, nsingle, nhi
        shift = [max(shift[d], s2[d]) for d in range(3)]
    return None


def lattice_brick_aggregates(xyz_nodes, xyz_free, dense_limit=128, passes=3, max_levels=13, owner=None, replicate_rows=150000, split=False):
    """The aggregates of -pc_type gamg on a scalar problem whose mesh nodes sit on a tensor-product lattice with strong
    couplings along every axis (the product: amg_bricks_level / k_lat_* in pfemfort_amd/csrc), restated from the COORDINATES
    alone: a node's position = ranks of its coordinates among the distinct values of ALL mesh nodes (``xyz_nodes`` [dim, nNode]);
    a level halves ``passes`` axes in turn -- x, y, z, x, ... skipping an axis without extent, the turn carried on to the next
    level --, an aggregate is a brick of positions, numbered in ascending (z, y, x) order of the occupied bricks; a brick's
    position on the next level is its brick coordinate.  ``xyz_free`` [dim, n]: the coordinates of the free nodes in dof order.
    Returns the list of aggregate maps, level by level, until a level has at most ``dense_limit`` dofs or stops shrinking by 20 %.

    ``owner`` [n] (one hierarchy across ranks; dofs numbered rank after rank): the rank that owns every dof.  Every rank's dofs
    must fill a box of positions (else None: the product leaves the brick path).  The planes where the owner changes are
    padded onto multiples of the level's brick size, on every level anew, so that no brick holds dofs of two owners; a rank
    numbers its own bricks (z, y, x), rank after rank.  From the first level of at most ``replicate_rows`` dofs (0: never) on,
    every rank holds the whole level: plain bricks again, of the positions with the padding closed up (rank among the occupied
    positions per axis).

    ``split`` (the product: amg_split_bricks): where some rank's dofs do NOT fill a box -- a METIS-like partition -- nothing is
    padded; a brick may hold dofs of several owners and every owner's part of it is an aggregate of its own, numbered by its
    owner in (z, y, x) order of the bricks, rank after rank.  (Partitions of boxes take the padded form either way.)"""
    xyz_nodes = np.atleast_2d(np.asarray(xyz_nodes, dtype=np.float64))
    xyz_free = np.atleast_2d(np.asarray(xyz_free, dtype=np.float64))
    dim, n = xyz_free.shape
    pos = np.zeros((3, n), np.int64)
    hi = [0, 0, 0]
    for d in range(dim):
        u = np.unique(xyz_nodes[d] + 0.0)
        pos[d] = np.searchsorted(u, xyz_free[d])
        assert np.array_equal(u[pos[d]], xyz_free[d])
        hi[d] = len(u) - 1
    aggs, axis = [], 0
    cuts = None
    if owner is not None:
        owner = np.asarray(owner, dtype=np.int64)
        assert len(owner) == n and (np.diff(owner) >= 0).all()
        boxes, split_now = [], False
        for q in np.unique(owner):
            sel = owner == q
            lo, up = pos[:, sel].min(axis=1), pos[:, sel].max(axis=1)
            if int(np.prod(up - lo + 1)) != int(sel.sum()):
                if not split:
                    return None
                split_now = True
            boxes.append((lo, up))
        if split_now:
            while n > dense_limit and len(aggs) + 1 < max_levels:
                shift, hi_c, axis_c = _lat_sim_passes(hi, axis, passes)
                if not any(shift):
                    break
                b = [pos[d] >> shift[d] for d in range(3)]
                nb = [(hi[d] >> shift[d]) + 1 for d in range(3)]
                lin = b[0] + nb[0] * (b[1] + nb[1] * b[2])
                nbt = nb[0] * nb[1] * nb[2]
                if owner is not None:
                    occ, agg = np.unique(owner * nbt + lin, return_inverse=True)
                    owner_c, occ = occ // nbt, occ % nbt
                else:
                    occ, agg = np.unique(lin, return_inverse=True)
                if len(occ) * 10 > n * 8:
                    break
                aggs.append(agg.astype(np.int64))
                pos = np.stack([occ % nb[0], (occ // nb[0]) % nb[1], occ // (nb[0] * nb[1])])
                n, hi, axis = len(occ), hi_c, axis_c
                if owner is not None:
                    owner = owner_c
                    if dense_limit < n <= replicate_rows:        # every rank holds the level whole from here on: positions closed up
                        owner = None
                        for d in range(3):
                            u = np.unique(pos[d])
                            pos[d] = np.searchsorted(u, pos[d])
                            hi[d] = len(u) - 1
            return aggs
        gmin = [min(b[0][d] for b in boxes) for d in range(3)]
        gmax = [max(b[1][d] for b in boxes) for d in range(3)]
        cuts = [sorted({0} | {int(b[0][d]) for b in boxes if b[0][d] > gmin[d]} | {int(b[1][d]) + 1 for b in boxes if b[1][d] < gmax[d]})
                for d in range(3)]
        padded = _lat_pad(cuts, [[], [], []], hi, (gmin, gmax), 0, passes)
        if padded is None:
            return None
        table, cuts, single, hi = padded
        pos = np.stack([table[d][pos[d]] for d in range(3)])
    while n > dense_limit and len(aggs) + 1 < max_levels:
        shift, hi_c, axis_c = _lat_sim_passes(hi, axis, passes)
        if not any(shift):
            break
        b = [pos[d] >> shift[d] for d in range(3)]
        nb = [(hi[d] >> shift[d]) + 1 for d in range(3)]
        lin = b[0] + nb[0] * (b[1] + nb[1] * b[2])
        if cuts is not None:          # distributed level: the owners number their own bricks, rank after rank
            nbt = nb[0] * nb[1] * nb[2]
            occ, agg = np.unique(owner * nbt + lin, return_inverse=True)
            owner_c, occ = occ // nbt, occ % nbt
        else:
            occ, agg = np.unique(lin, return_inverse=True)
        if len(occ) * 10 > n * 8:
            break
        aggs.append(agg.astype(np.int64))
        pos = np.stack([occ % nb[0], (occ // nb[0]) % nb[1], occ // (nb[0] * nb[1])])
        n, hi, axis = len(occ), hi_c, axis_c
        if cuts is not None:
            owner = owner_c
            padded = _lat_pad([[c >> shift[d] for c in cuts[d]] for d in range(3)], single, hi,
                              ([int(pos[d].min()) for d in range(3)], [int(pos[d].max()) for d in range(3)]), axis, passes)
            assert padded is not None
            table, cuts, single, hi = padded
            pos = np.stack([table[d][pos[d]] for d in range(3)])
            if dense_limit < n <= replicate_rows:
                cuts = None              # every rank holds the level whole from here on: the padding is closed up again --
                for d in range(3):       # a position becomes its rank among the occupied positions of its axis
                    u = np.unique(pos[d])
                    pos[d] = np.searchsorted(u, pos[d])
                    hi[d] = len(u) - 1
    return aggs


def lattice_by_numbering(conn, n_nodes):
    """A box's NUMBERING on a mesh of tetrahedra whose nodes need not sit on a lattice (the product: lattice_positions_by_numbering
    in pfemfort_amd/csrc/pfem_device.hip, k_latnum_check): the other nodes of every element lie at i + j a + k b from each of
    its nodes with i, j, k in {-1, 0, 1}; a and b are looked for among the (at most 13) distinct absolute differences themselves
    -- ascending pairs with b a multiple of a and a divisor of the node count, every axis in use -- and every element is held to
    the positions (n % a, (n % b) / a, n / b): its nodes within one cell of each other along every axis (a box of 6 x 5 x 4 nodes
    whose cells are cut along e_z - e_y offers b = 24 before b = 30: the first pair that passes this test is taken).
    ``conn`` [4, nElem].  Returns (a, b, positions [3, n_nodes]) or None."""
    conn = np.asarray(conn, dtype=np.int64)
    if conn.shape[0] != 4 or n_nodes < 8:
        return None
    diffs = np.unique(np.abs(conn[:, None, :] - conn[None, :, :]))
    diffs = diffs[diffs > 0]
    if len(diffs) == 0 or len(diffs) > 13:
        return None

    def fits(a, b):
        if a < 2 or b < 2 * a or b % a or n_nodes % b or a > 1024 or b // a > 1024 or n_nodes // b > 1024 or n_nodes // b < 2:
            return False
        axis = [False, False, False]
        for d in diffs:
            hit = [(i, j, k) for k in (0, 1) for j in (-1, 0, 1) for i in (-1, 0, 1) if i + j * a + k * b == d]
            if not hit:
                return False
            i, j, k = hit[0]
            axis[0] |= i != 0
            axis[1] |= j != 0
            axis[2] |= k != 0
        return all(axis)

    for p in range(len(diffs)):
        for q in range(p + 1, len(diffs)):
            a, b = int(diffs[p]), int(diffs[q])
            if not fits(a, b):
                continue
            pos = np.stack([np.arange(n_nodes) % a, (np.arange(n_nodes) % b) // a, np.arange(n_nodes) // b])
            pe = pos[:, conn]                              # [3, 4, nElem]
            if (pe.max(axis=1) - pe.min(axis=1)).max() <= 1:
                return a, b, pos
    return None


def _node_brick_axis(stretches, fine, halvings, first_max):
    """One axis of a node-brick level (pfem_amg.inc: node_brick_axis): the occupied positions come in stretches of one owner
    (inclusive ends; one rank: one stretch); inside a stretch f positions make a brick, aligned on multiples of f, a brick at either
    end that would hold at most f // 2 positions joined to its neighbour, ``halvings`` times over when f = 2 (level 0 of a
    displacement problem: f = 3 from an extent of 6 on, 4 from 24 on).  The first stretch keeps brick = position // f, every further
    stretch goes on where the one before it ended.  Returns (table position -> brick coordinate, the stretches in brick coordinates)."""
    table, out, base = np.zeros(1024, np.int64), [], 0
    for k, (s0, s1) in enumerate(stretches):
        lo, hi, m = s0, s1, np.arange(s0, s1 + 1)
        extent = s1 - s0 + 1
        want = 4 if extent >= 24 else (3 if extent >= 6 else 2)
        f = max(2, min(want, first_max, extent // 2)) if (fine and halvings > 0) else 2
        for _ in range(1 if f > 2 else halvings):
            bmin = lo // f + (1 if (f - lo % f) <= f // 2 else 0)
            bmax = hi // f - (1 if (hi % f + 1) <= f // 2 else 0)
            pairs = bmin <= bmax
            m = np.clip(m // f, bmin, bmax) if pairs else m // f
            lo, hi = (bmin, bmax) if pairs else (lo // f, hi // f)
        if k == 0:
            base = lo
        table[s0:s1 + 1] = m - lo + base
        out.append((base, base + hi - lo))
        base += hi - lo + 1
    return table, out


def lattice_node_brick_aggregates(xyz_nodes, xyz_free_nodes, dim=3, owner=None, first_max=4, passes=3, dense_limit=128, max_levels=13,
                                  replicate_rows=150000):
    """The NODE aggregates of -pc_type gamg on a displacement problem (rigid-body transfer) whose mesh nodes sit on a lattice whose
    lines are full (the product: amg_node_bricks / k_rbm_lat_* in pfemfort_amd/csrc), restated from the COORDINATES alone.
    ``xyz_nodes`` [dim, nNode]: all mesh nodes (a position = rank among the distinct values per axis); ``xyz_free_nodes`` [dim, n]:
    the free nodes in node order (dofs / dim).  A level halves ``passes`` axes in turn as the scalar bricks do; the bricks of an
    axis come from _node_brick_axis; aggregates = the bricks of the box, numbered in (z, y, x) order.  A coarse node has
    dim + (3 if dim == 3 else 1) dofs and sits at its brick coordinate.

    ``owner`` [n] (one hierarchy across ranks, nodes numbered rank after rank): every rank's nodes must fill a box of positions
    (else None); the bricks are cut where the owner changes, every rank numbers the bricks of its own box, rank after rank.  From
    the first level of at most ``replicate_rows`` dofs on every rank holds the whole level: positions = ranks among the occupied
    brick coordinates (what the centroids' distinct values give), one stretch per axis.
    Returns the list of node-aggregate maps, level by level."""
    xyz_nodes = np.atleast_2d(np.asarray(xyz_nodes, dtype=np.float64))
    xf = np.atleast_2d(np.asarray(xyz_free_nodes, dtype=np.float64))
    n = xf.shape[1]
    pos = np.zeros((3, n), np.int64)
    hi = [0, 0, 0]
    for d in range(xf.shape[0]):
        u = np.unique(xyz_nodes[d] + 0.0)
        pos[d] = np.searchsorted(u, xf[d])
        assert np.array_equal(u[pos[d]], xf[d])
        hi[d] = len(u) - 1
    cb = dim + (3 if dim == 3 else 1)
    bs = dim
    aggs, axis, fine = [], 0, True
    stretches = None
    if owner is not None:
        owner = np.asarray(owner, dtype=np.int64)
        assert len(owner) == n and (np.diff(owner) >= 0).all()
        boxes = []
        for q in np.unique(owner):
            sel = owner == q
            lo, up = pos[:, sel].min(axis=1), pos[:, sel].max(axis=1)
            if int(np.prod(up - lo + 1)) != int(sel.sum()):
                return None
            boxes.append((lo, up))
        stretches = []
        for d in range(3):
            gmin, gmax = min(b[0][d] for b in boxes), max(b[1][d] for b in boxes)
            cuts = {int(b[0][d]) for b in boxes if b[0][d] > gmin} | {int(b[1][d]) + 1 for b in boxes if b[1][d] < gmax}
            bnd = sorted({int(gmin), int(gmax) + 1} | {c for c in cuts if gmin < c <= gmax})
            stretches.append([(bnd[k], bnd[k + 1] - 1) for k in range(len(bnd) - 1)])
    elif int(np.prod(pos.max(axis=1) - pos.min(axis=1) + 1)) != n:
        return None
    while n * bs > dense_limit and len(aggs) + 1 < max_levels:
        shift, hi_c, axis_c = _lat_sim_passes(hi, axis, passes)
        if not any(shift):
            break
        b, nhi, new_str = [], list(hi_c), []
        for d in range(3):
            one = [(int(pos[d].min()), int(pos[d].max()))]
            table, out = _node_brick_axis(stretches[d] if stretches is not None else one, fine, shift[d], first_max)
            b.append(table[pos[d]])
            new_str.append(out)
            if stretches is not None:
                nhi[d] = out[-1][1]
            else:
                extent = one[0][1] - one[0][0] + 1
                want = 4 if extent >= 24 else (3 if extent >= 6 else 2)
                f = max(2, min(want, first_max, extent // 2)) if (fine and shift[d] > 0) else 2
                if f > 2:
                    nhi[d] = hi[d] // f
        b = np.stack(b)
        if stretches is not None:        # every owner numbers the bricks of its own box (z, y, x), rank after rank
            agg = np.zeros(n, np.int64)
            off, owner_c, pos_c = 0, [], []
            for q in np.unique(owner):
                sel = owner == q
                lo, up = b[:, sel].min(axis=1), b[:, sel].max(axis=1)
                nbk = up - lo + 1
                agg[sel] = off + (b[0, sel] - lo[0]) + nbk[0] * ((b[1, sel] - lo[1]) + nbk[1] * (b[2, sel] - lo[2]))
                cnt = int(np.prod(nbk))
                a = np.arange(cnt)
                pos_c.append(np.stack([a % nbk[0] + lo[0], (a // nbk[0]) % nbk[1] + lo[1], a // (nbk[0] * nbk[1]) + lo[2]]))
                owner_c.append(np.full(cnt, q))
                off += cnt
            na, pos_c, owner_c = off, np.concatenate(pos_c, axis=1), np.concatenate(owner_c)
        else:
            lo, up = b.min(axis=1), b.max(axis=1)
            nbk = up - lo + 1
            agg = (b[0] - lo[0]) + nbk[0] * ((b[1] - lo[1]) + nbk[1] * (b[2] - lo[2]))
            na = int(np.prod(nbk))
            a = np.arange(na)
            pos_c, owner_c = np.stack([a % nbk[0] + lo[0], (a // nbk[0]) % nbk[1] + lo[1], a // (nbk[0] * nbk[1]) + lo[2]]), None
        if na < 1 or na * 10 > n * 8 or na * cb * 10 > n * bs * 8:
            break
        aggs.append(agg)
        pos, n, hi, axis, bs, fine, owner = pos_c, na, nhi, axis_c, cb, False, owner_c
        if stretches is not None:
            stretches = new_str
            if dense_limit < n * bs <= replicate_rows:      # every rank holds the level whole from here on
                stretches, owner = None, None
                for d in range(3):
                    u = np.unique(pos[d])
                    pos[d] = np.searchsorted(u, pos[d])
                    hi[d] = len(u) - 1
    return aggs


def rbm_prolongator(node_agg, xyz, dim, fine_bs):
    """Tentative prolongator with the rigid-body modes of every aggregate (the product: pfem_amg_rbm.hpp; PETSc reaches the
    same coarse space through MatSetNearNullSpace / PCSetCoordinates ahead of PCGAMG, tetraelasticityparallelimpl1.F:894-902).
    ``node_agg[i]`` = aggregate of node i, ``xyz`` [3, n_nodes] the nodes' coordinates, nodes of ``fine_bs`` dofs (dim
    displacements, then -- below the assembled matrix -- the rotations).  A coarse node has dim translations T and 3 (plane: 1)
    rotations W about the aggregate's centroid:  u_i = T + W x r_i,  w_i = W.  Aggregates whose nodes do not span enough
    space for the rotations to be independent (collinear in space, single in the plane; assembled matrix only) keep their
    translations: their offsets are 0.  Returns (P as scipy CSR, centroids [3, n_aggregates])."""
    import scipy.sparse as sp
    node_agg = np.asarray(node_agg, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64).reshape(3, -1)
    nn, na = len(node_agg), int(node_agg.max()) + 1
    nr = 3 if dim == 3 else 1
    cb = dim + nr
    cnt = np.bincount(node_agg, minlength=na).astype(np.float64)
    cen = np.stack([np.bincount(node_agg, weights=xyz[d], minlength=na) / cnt for d in range(3)])
    r = xyz - cen[:, node_agg]
    if fine_bs == dim:
        S = {(a, b): np.bincount(node_agg, weights=r[a] * r[b], minlength=na) for a in range(3) for b in range(a, 3)}
        tr = S[0, 0] + S[1, 1] + S[2, 2]
        if dim == 3:
            m2 = (S[0, 0] * S[1, 1] - S[0, 1] ** 2) + (S[0, 0] * S[2, 2] - S[0, 2] ** 2) + (S[1, 1] * S[2, 2] - S[1, 2] ** 2)
            ok = (tr > 0.0) & (m2 > 1e-8 * tr * tr)
        else:
            ok = tr > 0.0
        r = r * ok[node_agg]
    node = np.arange(nn)
    rows, cols_, vals_ = [], [], []

    def put(c, a, v):
        rows.append(node * fine_bs + c); cols_.append(node_agg * cb + a); vals_.append(np.broadcast_to(v, (nn,)).astype(np.float64))
    for c in range(dim):
        put(c, c, 1.0)
    if dim == 3:          # u = W x r
        put(0, 4, r[2]); put(0, 5, -r[1])
        put(1, 5, r[0]); put(1, 3, -r[2])
        put(2, 3, r[1]); put(2, 4, -r[0])
    else:
        put(0, 2, -r[1]); put(1, 2, r[0])
    if fine_bs > dim:
        for k in range(nr):
            put(dim + k, dim + k, 1.0)
    P = sp.csr_matrix((np.concatenate(vals_), (np.concatenate(rows), np.concatenate(cols_))), shape=(nn * fine_bs, na * cb))
    return P, cen


def amg_cycle(rowptr, cols, vals, aggregates, cheb_degree=2, eig_ratio=8.0, coarse_scale=1.5, dense_limit=128, coarsest_sweeps=8,
              fine_degree=1, lam_given=None, gamma=1, gamma_from=1, gamma_to=99):
    """z = M^-1 r of the product's -pc_type gamg (pfemfort_amd/csrc/pfem_amg.inc), restated in numpy / scipy.sparse GIVEN the
    aggregates (``aggregates[l][i]`` = coarse dof of dof i of level l; the product forms them by pairwise matching and
    hands them over for this check; an entry may also be a prolongator itself, see rbm_prolongator).  Everything else is restated: piecewise-constant prolongation P, Galerkin operators
    P^T A P, Chebyshev smoothing of degree ``cheb_degree`` (``fine_degree`` on the matrix itself) on D^-1 A over [lmax/eig_ratio, lmax] with the Gershgorin bound
    lmax = max_i sum_j |a_ij| / a_ii, one symmetric V(1,1) cycle with the coarse correction scaled by ``coarse_scale``, a
    dense solve on the last level when it has at most ``dense_limit`` rows (else Chebyshev of degree ``coarsest_sweeps``).
    ``gamma`` = 2: W-cycle (-pc_mg_cycle_type w) -- the coarse problem of every level l with gamma_from <= l <= gamma_to that is
    not the last is visited twice, the second time on the residual of the first, the two answers added.
    Returns the function r -> z."""
    import scipy.sparse as sp
    N = len(rowptr) - 1
    A = sp.csr_matrix((np.asarray(vals, dtype=np.float64), np.asarray(cols), np.asarray(rowptr)), shape=(N, N))
    levels = [A]
    P = []
    rbm = False
    for agg in aggregates:
        if sp.issparse(agg):          # an explicit prolongator (rbm_prolongator): Galerkin product with it; the rotation dofs of an
            Pl = agg.tocsr()          # aggregate without extent are idle dofs with a unit diagonal
            rbm = True
            Ac = (Pl.T @ levels[-1] @ Pl).tocsr()
            dg = Ac.diagonal()
            if (dg == 0.0).any():
                Ac = (Ac + sp.diags((dg == 0.0).astype(np.float64))).tocsr()
        else:
            agg = np.asarray(agg, dtype=np.int64)
            nc = int(agg.max()) + 1
            Pl = sp.csr_matrix((np.ones(len(agg)), (np.arange(len(agg)), agg)), shape=(len(agg), nc))
            Ac = (Pl.T @ levels[-1] @ Pl).tocsr()
        P.append(Pl)
        levels.append(Ac)
    dinv, lam = [], []
    for Al in levels:
        d = Al.diagonal()
        dinv.append(1.0 / d)
        lam.append(float((abs(Al) @ np.ones(Al.shape[0]) / d).max()))
        if rbm:     # second bound, on D^-1/2 A D^-1/2 (independent of the scaling of the rotation dofs); the smaller one counts
            sq = 1.0 / np.sqrt(d)
            lam[-1] = min(lam[-1], float(((abs(Al) @ sq) * sq).max()))
    lam_true = list(lam)
    if lam_given is not None:
        # one hierarchy across several ranks: the product sums the absolute values of the RANKS' SHARES of an interface
        # entry, an upper bound of the absolute value of the entry (the caller checks lam_true <= lam_given)
        lam = [float(v) for v in lam_given]
    dense = len(levels) > 1 and levels[-1].shape[0] <= dense_limit
    Ainv = np.linalg.inv(levels[-1].toarray()) if dense else None

    def smooth(l, x, rhs, deg):
        Al, d = levels[l], dinv[l]
        lmax = lam[l]; lmin = lmax / eig_ratio
        theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        r = rhs.copy() if x is None else rhs - Al @ x
        dd = d * r / theta
        x = dd.copy() if x is None else x + dd
        for _ in range(1, deg):
            r = r - Al @ dd
            rho_new = 1.0 / (2.0 * sigma - rho)
            dd = rho_new * rho * dd + (2.0 * rho_new / delta) * (d * r)
            x = x + dd
            rho = rho_new
        return x

    def cycle(l, rhs):
        if l == len(levels) - 1:
            if dense:
                return Ainv @ rhs
            return smooth(l, None, rhs, cheb_degree if len(levels) == 1 else coarsest_sweeps)
        deg = fine_degree if (l == 0 and fine_degree) else cheb_degree
        x = smooth(l, None, rhs, deg)
        rc = P[l].T @ (rhs - levels[l] @ x)
        xc = cycle(l + 1, rc)
        for _ in range(1, gamma if (gamma_from <= l + 1 <= gamma_to and l + 2 < len(levels)) else 1):      # W-cycle: the coarse problem gets a second go
            xc = xc + cycle(l + 1, rc - levels[l + 1] @ xc)
        x = x + coarse_scale * (P[l] @ xc)
        return smooth(l, x, rhs, deg)

    apply = lambda r: cycle(0, r)          # noqa: E731
    apply.lam_true = lam_true
    return apply


def pcg_with(rowptr, cols, vals, b, M, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000):
    """CG with PETSc's KSPCG semantics (zero initial guess, preconditioned norm, test after the update)
    and ANY symmetric positive definite preconditioner ``M``: r -> z.  Returns (x, its, reason, rnorm, history)."""
    import scipy.sparse as sp
    N = len(rowptr) - 1
    A = sp.csr_matrix((np.asarray(vals, dtype=np.float64), np.asarray(cols), np.asarray(rowptr)), shape=(N, N))
    b = _f64(b)
    x = np.zeros(N)
    r = b.copy()
    z = M(r)
    beta = float(r @ z)
    rn0 = float(np.sqrt(z @ z))
    hist = [rn0]
    if rn0 <= abstol:
        return x, 0, 3, rn0, np.array(hist)
    ttol = max(rtol * rn0, abstol)
    p = z.copy()
    for it in range(1, maxits + 1):
        w = A @ p
        pw = float(p @ w)
        if not pw > 0.0:
            return x, it - 1, -10, hist[-1], np.array(hist)
        alpha = beta / pw
        x += alpha * p
        r -= alpha * w
        z = M(r)
        bn = float(r @ z)
        rn = float(np.sqrt(z @ z))
        hist.append(rn)
        if rn <= ttol:
            return x, it, 2, rn, np.array(hist)
        if rn >= dtol * rn0:
            return x, it, -4, rn, np.array(hist)
        if bn < 0.0:
            return x, it, -8, rn, np.array(hist)
        p = z + (bn / beta) * p
        beta = bn
    return x, maxits, -3, hist[-1], np.array(hist)


def pcg_with_single_reduction(rowptr, cols, vals, b, M, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000):
    """The single-reduction form of pcg_with (PETSc: KSPCGUseSingleReduction; the product: k_pcg1_step): s = A z, (p,Ap) by the
    recurrence (z,s) - beta^2 (p,Ap)_old, so that (z,s), (r,z), (z,z) are reduced together; step k judges iterate k from
    ||z_k|| before advancing.  Same iterates as pcg_with up to rounding.  Returns (x, its, reason, rnorm, history)."""
    import scipy.sparse as sp
    N = len(rowptr) - 1
    A = sp.csr_matrix((np.asarray(vals, dtype=np.float64), np.asarray(cols), np.asarray(rowptr)), shape=(N, N))
    b = _f64(b)
    x = np.zeros(N)
    r = b.copy()
    p = np.zeros(N)
    w = np.zeros(N)
    hist = []
    beta_old = dpi_old = 0.0
    rn0 = ttol = 0.0
    for it in range(0, maxits + 2):
        z = M(r)
        s = A @ z
        zs, rz, zz = float(z @ s), float(r @ z), float(z @ z)
        rn = float(np.sqrt(zz))
        hist.append(rn)
        if it == 0:
            rn0, ttol = rn, max(rtol * rn, abstol)
            if rn <= abstol:
                return x, 0, 3, rn, np.array(hist)
        elif rn <= ttol:
            return x, it, 2, rn, np.array(hist)
        elif rn >= dtol * rn0:
            return x, it, -4, rn, np.array(hist)
        if rz < 0.0:
            return x, it, -8, rn, np.array(hist)
        if it >= maxits:
            return x, it, -3, rn, np.array(hist)
        if it == 0:
            beta, dpi = 0.0, zs
        else:
            beta = rz / beta_old
            dpi = zs - rz * rz * dpi_old / (beta_old * beta_old)
        if not dpi > 0.0:
            return x, it + 1, -10, rn, np.array(hist)
        alpha = rz / dpi
        p = z + beta * p
        w = s + beta * w
        x += alpha * p
        r -= alpha * w
        beta_old, dpi_old = rz, dpi
    return x, maxits, -3, hist[-1], np.array(hist)


def pcg_amg(rowptr, cols, vals, b, aggregates, cheb_degree=2, eig_ratio=8.0, coarse_scale=1.5, rtol=1e-5, abstol=1e-50, dtol=1e5,
            maxits=10000, dense_limit=128, coarsest_sweeps=8, fine_degree=1, lam_given=None, lam_true_out=None, single_reduction=False,
            gamma=1, gamma_to=99):
    """CG preconditioned by one V(1,1) -- gamma = 2: W(1,1) down to level gamma_to -- cycle of plain-aggregation multigrid on the
    whole matrix: amg_cycle + pcg_with."""
    M = amg_cycle(rowptr, cols, vals, aggregates, cheb_degree, eig_ratio, coarse_scale, dense_limit, coarsest_sweeps, fine_degree, lam_given,
                  gamma, 1, gamma_to)
    if lam_true_out is not None:
        lam_true_out[:] = M.lam_true
    return (pcg_with_single_reduction if single_reduction else pcg_with)(rowptr, cols, vals, b, M, rtol, abstol, dtol, maxits)


def pcg_bjacobi_amg(rowptr, cols, vals, b, blocks, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000, **amg):
    """The product's multi-rank form of -pc_type gamg: block Jacobi over the ranks' row blocks, every block preconditioned by
    the V-cycle of ITS OWN matrix -- ``blocks`` = [(row_start, (rowptr_r, cols_r, vals_r), aggregates_r), ...], where the block
    matrix is the rank's owned diagonal block as the rank assembled it from its own elements (the interface rows lack the
    neighbour's share; PETSc's PCBJACOBI would use the assembled block).  CG on the assembled global matrix."""
    cyc = [(rs, len(m[0]) - 1, amg_cycle(*m, aggs, **amg)) for rs, m, aggs in blocks if len(m[0]) > 1]      # an idle rank has no block

    def M(r):
        z = np.zeros_like(r)
        for rs, nr, c in cyc:
            if nr:
                z[rs:rs + nr] = c(r[rs:rs + nr])
        return z
    return pcg_with(rowptr, cols, vals, b, M, rtol, abstol, dtol, maxits)


def row_groups(rowptr, cols, max_rows=3):
    """First rows of the row groups a node-block preconditioner works on: consecutive rows with identical
    column sets (the dof rows of a node), at most ``max_rows`` per group; last entry = number of rows."""
    N = len(rowptr) - 1
    starts = []
    run0 = 0
    for r in range(N):
        brk = r == 0 or (rowptr[r + 1] - rowptr[r]) != (rowptr[r] - rowptr[r - 1]) or rowptr[r + 1] == rowptr[r] or \
            not np.array_equal(cols[rowptr[r]:rowptr[r + 1]], cols[rowptr[r - 1]:rowptr[r]])
        if brk:
            run0 = r
        if (r - run0) % max_rows == 0:
            starts.append(r)
    return np.array(starts + [N], np.int64)


def pcg_block_jacobi(rowptr, cols, vals, b, groups, rtol=1e-5, abstol=1e-50, dtol=1e5, maxits=10000):
    """PETSc's CG with -pc_type pbjacobi restated in numpy (test sizes only): M = the diagonal blocks of A
    over ``groups``; same preconditioned-norm convergence test as pcg_jacobi (KSPConvergedDefault)."""
    import scipy.sparse as sp
    N = len(rowptr) - 1
    A = sp.csr_matrix((vals, cols, rowptr), shape=(N, N))
    inv = []
    for g in range(len(groups) - 1):
        i0, i1 = int(groups[g]), int(groups[g + 1])
        inv.append(np.linalg.inv(A[i0:i1, i0:i1].toarray()))

    def apply(r):
        z = np.empty_like(r)
        for g, Bi in enumerate(inv):
            i0, i1 = int(groups[g]), int(groups[g + 1])
            z[i0:i1] = Bi @ r[i0:i1]
        return z

    x = np.zeros(N); r = np.asarray(b, float).copy(); z = apply(r); pv = z.copy()
    beta = float(r @ z); rn0 = float(np.sqrt(z @ z)); ttol = max(rtol * rn0, abstol)
    if rn0 <= abstol:
        return x, 0, 3, rn0
    for it in range(1, maxits + 1):
        w = A @ pv
        pw = float(pv @ w)
        if not pw > 0.0:
            return x, it - 1, -10, rn0
        alpha = beta / pw
        x += alpha * pv; r -= alpha * w
        z = apply(r)
        bnew = float(r @ z); rn = float(np.sqrt(z @ z))
        if rn <= ttol:
            return x, it, 2, rn
        if rn >= dtol * rn0:
            return x, it, -4, rn
        pv = z + (bnew / beta) * pv
        beta = bnew
    return x, maxits, -3, rn


# ----------------------------------------------------------------------------
# the whole path, as the driver runs it on one or several ranks
# ----------------------------------------------------------------------------
@dataclass
class Problem:
    kind: int
    mesh: Mesh
    dm: DofMap
    xyz_new: np.ndarray
    conn_new: np.ndarray
    edof: np.ndarray
    rowptr: np.ndarray
    cols: np.ndarray
    vals: np.ndarray
    rhs: np.ndarray
    elemData: np.ndarray


def setup_problem(kind, mesh: Mesh, elemData=None, nParts=1, node_proc_id=None) -> Problem:
    """tetrapoissonparallelimpl1.F:316-884 on one process (all elements assembled)."""
    if elemData is None:
        elemData = {ELAST_TET: ELAST_ELEMDATA, ELAST_TRIA: ELAST2D_ELEMDATA}.get(kind, POISSON_ELEMDATA)
    ndof = NDOF[kind]
    dm = dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, nParts, node_proc_id)
    conn_new = dm.node_map_get_new[mesh.conn].astype(np.int32)
    xyz_new = np.ascontiguousarray(mesh.xyz[:, dm.node_map_get_old])
    edof = elem_dof_array(conn_new, dm.NodeDofArrayNew)
    rowptr, cols = csr_pattern(edof, dm.size_global)
    vals, rhs = assemble(kind, xyz_new, conn_new, edof, dm.solnApplied, elemData, dm.size_global, rowptr, cols)
    return Problem(kind, mesh, dm, xyz_new, conn_new, edof, rowptr, cols, vals, rhs, np.asarray(elemData))


def full_solution(prob: Problem, u_free):
    """solnVTK of the driver (:914-941): values by OLD node id, (nNode, ndof)."""
    ndof = NDOF[prob.kind]
    dm = prob.dm
    full = dm.solnApplied.reshape(-1, ndof).copy()           # NEW order, Dirichlet values
    assy = assy_for_soln(dm.NodeDofArrayNew)
    full.reshape(-1)[assy] = u_free
    out = np.empty_like(full)
    out[dm.node_map_get_old] = full
    return out
