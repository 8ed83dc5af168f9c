"""Host-side bookkeeping of the PFEMFort drivers, through the C ABI (csrc/pfem_host.cpp).

numpy wrappers of the integer/mesh logic the drivers run before the element loop:
mesh files, the structured generator (genTetra.cpp), Dirichlet/DOF
numbering and partition renumbering (tetrapoissonparallelimpl1.F:316-367, 393-734) and
the per-element compat routines (MODULE ElementUtilitiesPoisson / ...Elasticity3D).
Arrays are column-major like the Fortran: ``xyz (ndim, nNode)``, ``conn (npElem, nElem)``,
``edof (nsize, nElem)``, 0-based, -1 = Dirichlet.
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
from dataclasses import dataclass

import numpy as np

from . import _lib as L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# REAL(4) literals of the drivers, widened to double (the reference is built without
# -fdefault-real-8): tetrapoissonparallelimpl1.F:822-823, tetraelasticityparallelimpl1.F:895-902
def _f32(v):
    return float(np.float32(v))


POISSON_ELEMDATA = np.array([1.0, 1.0, 1.0])
ELAST_ELEMDATA = np.array([_f32(240.565), _f32(0.3), 1.0, _f32(0.1), 0.0, 0.0])
TIMEDATA = np.array([0.0, 1.0, 0.0])
# triaelasticityparallelimpl1.F:907 sets only E and nu; thick / bforce are read uninitialised there
# (SURVEY 8f.1): the intended unit thickness and zero body force are used
ELAST2D_ELEMDATA = np.array([_f32(240.565), _f32(0.3), 1.0, 0.0, 0.0])


@dataclass
class Mesh:
    xyz: np.ndarray        # (ndim, nNode)
    conn: np.ndarray       # (npElem, nElem) int32, 0-based, OLD numbering
    bc_node: np.ndarray    # (nDBC,) int32 0-based
    bc_dof: np.ndarray     # (nDBC,) int32 0-based
    bc_val: np.ndarray     # (nDBC,)
    box: tuple | None = None   # (nEx, nEy, nEz) for generated meshes
    force_node: np.ndarray | None = None   # ForceBC file: node (0-based), dof (0-based), value
    force_dof: np.ndarray | None = None
    force_val: np.ndarray | None = None

    @property
    def nNode(self):
        return self.xyz.shape[1]

    @property
    def nElem(self):
        return self.conn.shape[1]


def read_table(path: str) -> np.ndarray:
    """One mesh file -> (rows, cols) float64, parsed by the library's multithreaded ASCII reader."""
    with (gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")) as f:
        buf = f.read()
    rows = C.c_int64(0); cols = C.c_int(0)
    L.check(L.lib().pfem_text_table_shape(buf, len(buf), C.byref(rows), C.byref(cols)), f"pfem_text_table_shape({path})")
    out = np.empty((cols.value, rows.value))
    if rows.value == 0:
        return out.T
    L.check(L.lib().pfem_text_table_parse(buf, len(buf), rows.value, cols.value, _p(out)), f"pfem_text_table_parse({path})")
    return out.T


def read_mesh(prefix: str) -> Mesh:
    """``<prefix>-nodes.dat[.gz]``, ``-elems``, ``-DirichBC``: whitespace ASCII, 1-based
    (tetrapoissonparallelimpl1.F:216-355)."""
    def load(kind):
        for ext in (".dat.gz", ".dat"):
            path = f"{prefix}-{kind}{ext}"
            if os.path.exists(path):
                return read_table(path)
        raise FileNotFoundError(f"{prefix}-{kind}.dat[.gz]")
    nodes, elems, bcs = load("nodes"), load("elems"), load("DirichBC")
    # read_table hands back the transpose of a (columns, records) array: slice THAT, so every step below is a
    # contiguous pass (the other order costs seconds at millions of elements)
    conn = elems.T[1:].astype(np.int32)
    conn -= 1
    m = Mesh(np.ascontiguousarray(nodes.T[1:]), conn,
             (bcs[:, 0] - 1).astype(np.int32), (bcs[:, 1] - 1).astype(np.int32), bcs[:, 2].copy())
    try:                                    # optional 4th file (tetraelasticityparallelimpl1.F:207-214)
        fb = load("ForceBC")
        m.force_node, m.force_dof, m.force_val = (fb[:, 0] - 1).astype(np.int32), (fb[:, 1] - 1).astype(np.int32), fb[:, 2].copy()
    except FileNotFoundError:
        pass
    return m


def gen_box_tets(x0, x1, nEx, y0, y1, nEy, z0, z1, nEz, bc_mode=0, ndof=1, kz=None, nodes=True) -> Mesh:
    """genTetra.cpp: structured box, 6 tets per hex.  ``kz=(k0,k1)`` emits only the elements of
    hex layers [k0,k1) (one rank's slab); nodes always cover the full grid."""
    k0, k1 = (0, nEz) if kz is None else kz
    nNode = (nEx + 1) * (nEy + 1) * (nEz + 1)
    nElem = 6 * nEx * nEy * (k1 - k0)
    xyz = np.empty((3, nNode)) if nodes else None
    conn = np.empty((4, nElem), dtype=np.int32)
    n = C.c_int64(0)
    head = (x0, x1, nEx, y0, y1, nEy, z0, z1, nEz, k0, k1, bc_mode, ndof)
    L.check(L.lib().pfem_gen_box_tets(*head, None, None, C.byref(n), None, None, None), "pfem_gen_box_tets")
    bn = np.empty(n.value, np.int32); bd = np.empty(n.value, np.int32); bv = np.empty(n.value)
    L.check(L.lib().pfem_gen_box_tets(*head, _p(xyz), _p(conn), C.byref(n), _p(bn), _p(bd), _p(bv)), "pfem_gen_box_tets")
    return Mesh(xyz, conn, bn, bd, bv, box=(nEx, nEy, nEz))


def box_slab_sizes(nEx, nEy, nEz, bc_mode=0, ndof=1, nparts=1, part=0, axis=2):
    """Closed-form sizes of slab ``part`` of the generated box in the reference's numbering (pfem_box_slab_sizes_axis).
    ``axis``: 0 x, 1 y, 2 z (default), -1: the axis with the most hex layers; the dict also names the axis taken and the
    slab's hex layers."""
    v = [C.c_int64(0) for _ in range(5)]
    w = [C.c_int(0) for _ in range(3)]
    L.check(L.lib().pfem_box_slab_sizes_axis(nEx, nEy, nEz, bc_mode, ndof, axis, nparts, part, *[C.byref(x) for x in v + w]),
            "pfem_box_slab_sizes_axis")
    keys = ("size_global", "row_start", "size_local", "nNode_local", "nElem_local", "axis", "layer0", "layer1")
    return {k: x.value for k, x in zip(keys, v + w)}


def partition_box_slabs(nEx, nEy, nEz, nParts, elements=True, axis=2):
    """Deterministic stand-in for METIS_PartMeshNodal (:464) on generated boxes: slabs of hex layers along ``axis``
    (-1: the longest).  ``elements=False`` skips the (large) elem_proc_id array and returns None for it."""
    epid = np.empty(6 * nEx * nEy * nEz, np.int32) if elements else None
    npid = np.empty((nEx + 1) * (nEy + 1) * (nEz + 1), np.int32)
    L.check(L.lib().pfem_partition_box_slabs_axis(nEx, nEy, nEz, axis, nParts, _p(epid), _p(npid)), "pfem_partition_box_slabs_axis")
    return epid, npid


def partition_rcb(mesh, nParts):
    """(elem_proc_id, node_proc_id) by recursive coordinate bisection of the element centroids: a geometric stand-in
    for METIS_PartMeshNodal (:464) on any mesh with coordinates."""
    xyz = _f64(mesh.xyz); conn = _i32(mesh.conn)
    epid = np.empty(conn.shape[1], np.int32); npid = np.empty(xyz.shape[1], np.int32)
    L.check(L.lib().pfem_partition_rcb(xyz.shape[1], xyz.shape[0], _p(xyz), conn.shape[1], conn.shape[0], _p(conn), nParts,
                                       _p(epid), _p(npid)), "pfem_partition_rcb")
    return epid, npid


def read_metis_partition(prefix: str, nParts: int):
    """(elem_proc_id, node_proc_id) from the files METIS' ``mpmetis <mesh> nParts`` writes
    (``<prefix>.epart.<nParts>`` / ``<prefix>.npart.<nParts>``, one 0-based part per line): the
    file hook for ``METIS_PartMeshNodal`` (tetrapoissonparallelimpl1.F:464), which is not installed here."""
    out = []
    for kind in ("epart", "npart"):
        t = read_table(f"{prefix}.{kind}.{nParts}")
        if t.ndim != 2 or t.shape[1] != 1:
            raise ValueError(f"{prefix}.{kind}.{nParts}: expected one part id per line")
        ids = t[:, 0].astype(np.int32)
        if len(ids) and (ids.min() < 0 or ids.max() >= nParts or np.any(ids != t[:, 0])):
            raise ValueError(f"{prefix}.{kind}.{nParts}: part id out of range")
        out.append(ids)
    return out[0], out[1]


@dataclass
class DofMap:
    node_map_get_old: np.ndarray
    node_map_get_new: np.ndarray
    NodeDofArrayNew: np.ndarray     # (nNode, ndof) 0-based ids, -1 = Dirichlet
    solnApplied: np.ndarray         # (nNode*ndof,) by NEW node*ndof+d
    node_start: np.ndarray
    node_end: np.ndarray
    row_start: np.ndarray
    row_end: np.ndarray
    size_global: int


def dof_numbering(nNode, ndof, bc_node, bc_dof, bc_val, nParts=1, node_proc_id=None) -> DofMap:
    old = np.empty(nNode, np.int32); new = np.empty(nNode, np.int32)
    nda = np.empty((nNode, ndof), np.int32); sa = np.empty(nNode * ndof)
    ns, ne, rs, re = (np.zeros(nParts, np.int64) for _ in range(4))
    sg = C.c_int64(0)
    npid = None if node_proc_id is None else _i32(node_proc_id)
    L.check(L.lib().pfem_dof_numbering(nNode, ndof, len(bc_node), _p(_i32(bc_node)), _p(_i32(bc_dof)), _p(_f64(bc_val)),
                                       nParts, _p(npid), _p(old), _p(new), _p(nda), _p(sa), _p(ns), _p(ne), _p(rs),
                                       _p(re), C.byref(sg)), "pfem_dof_numbering")
    return DofMap(old, new, nda, sa, ns, ne, rs, re, sg.value)


def renumber_mesh(mesh, dm: DofMap):
    """Connectivity and coordinates in the NEW node numbering: ``elemNodeConn = node_map_get_new(.)``
    (tetrapoissonparallelimpl1.F:659-664) and ``coords(node_map_get_old(.))`` (:832-838)."""
    conn = _i32(mesh.conn); xyz = _f64(mesh.xyz)
    conn_new = np.empty(conn.shape, np.int32)
    xyz_new = np.empty(xyz.shape, np.float64)
    L.check(L.lib().pfem_renumber_mesh(xyz.shape[1], xyz.shape[0], conn.shape[1], conn.shape[0], _p(conn), _p(xyz),
                                       _p(_i32(dm.node_map_get_new)), _p(_i32(dm.node_map_get_old)), _p(conn_new),
                                       _p(xyz_new)), "pfem_renumber_mesh")
    return conn_new, xyz_new


def elem_dof_array(conn_new, NodeDofArrayNew):
    conn_new = _i32(conn_new)
    npE, nElem = conn_new.shape
    ndof = NodeDofArrayNew.shape[1]
    edof = np.empty((npE * ndof, nElem), np.int32)
    L.check(L.lib().pfem_elem_dof_array(nElem, npE, ndof, _p(conn_new), _p(_i32(NodeDofArrayNew)), _p(edof)),
            "pfem_elem_dof_array")
    return edof


def assy_for_soln(NodeDofArrayNew):
    nNode, ndof = NodeDofArrayNew.shape
    out = np.empty(int((NodeDofArrayNew >= 0).sum()), np.int32)
    L.check(L.lib().pfem_assy_for_soln(nNode, ndof, _p(_i32(NodeDofArrayNew)), _p(out)), "pfem_assy_for_soln")
    return out


# ---- per-element compat routines (one element per call, like the Fortran) -----------
def StiffnessResidualPoissonLinearTetra(xNode, yNode, zNode, elemData, timeData, valC, valDotC=None):
    """elementutilitiespoisson.F:107; returns (Klocal[4,4], Flocal[4])."""
    K = np.empty((4, 4), order="F"); F = np.empty(4)
    L.check(L.lib().pfem_poisson_tet_ke(_p(_f64(xNode)), _p(_f64(yNode)), _p(_f64(zNode)), _p(_f64(elemData)),
                                        _p(_f64(timeData)), _p(_f64(valC)), _p(K), _p(F)),
            "StiffnessResidualPoissonLinearTetra")
    return K, F


def StiffnessResidualPoissonLinearTria(xNode, yNode, elemData, timeData, valC, valDotC=None):
    """elementutilitiespoisson.F:23; returns (Klocal[3,3], Flocal[3])."""
    K = np.empty((3, 3), order="F"); F = np.empty(3)
    L.check(L.lib().pfem_poisson_tria_ke(_p(_f64(xNode)), _p(_f64(yNode)), _p(_f64(elemData)), _p(_f64(timeData)),
                                         _p(_f64(valC)), _p(K), _p(F)), "StiffnessResidualPoissonLinearTria")
    return K, F


def StiffnessResidualElasticityLinearTetra(xNode, yNode, zNode, elemData, timeData, valC, valDotC=None):
    """elementutilitieselasticity3D.F:248 (intended semantics); returns (Klocal[12,12], Flocal[12])."""
    K = np.empty((12, 12), order="F"); F = np.empty(12)
    L.check(L.lib().pfem_elast_tet_ke(_p(_f64(xNode)), _p(_f64(yNode)), _p(_f64(zNode)), _p(_f64(elemData)),
                                      _p(_f64(timeData)), _p(_f64(valC)), _p(K), _p(F)),
            "StiffnessResidualElasticityLinearTetra")
    return K, F


def StiffnessResidualElasticityLinearTria(xNode, yNode, elemData, timeData, valC, valDotC=None):
    """elementutilitieselasticity2D.F:23; returns (Klocal[6,6], Flocal[6])."""
    K = np.empty((6, 6), order="F"); F = np.empty(6)
    L.check(L.lib().pfem_elast_tria_ke(_p(_f64(xNode)), _p(_f64(yNode)), _p(_f64(elemData)), _p(_f64(timeData)),
                                       _p(_f64(valC)), _p(K), _p(F)), "StiffnessResidualElasticityLinearTria")
    return K, F


def find_ghosts(edof, row_start, n_owned):
    """Ascending unique global dof ids of ``edof`` outside the owned block (host-only)."""
    e = _i32(edof).ravel()
    n = C.c_int64(0)
    L.check(L.lib().pfem_find_ghosts(e.size, _p(e), row_start, n_owned, C.byref(n), None), "pfem_find_ghosts")
    g = np.empty(n.value, np.int64)
    L.check(L.lib().pfem_find_ghosts(e.size, _p(e), row_start, n_owned, C.byref(n), _p(g)), "pfem_find_ghosts")
    return g


def neighbour_plan(rank, row_ranges, ghost_lists):
    """The neighbour plan of ``rank`` (include/pfem_amd.h, section 5; host-only integer logic):
    ``row_ranges[r] = (row_start, row_end)`` and ``ghost_lists[r]`` = ascending global ghost dof ids of rank r.
    Returns (peers int32[n], peer_off int64[n+1], shared_gid int64[peer_off[-1]])."""
    world = len(row_ranges)
    rs = np.ascontiguousarray([r[0] for r in row_ranges], dtype=np.int64)
    re = np.ascontiguousarray([r[1] for r in row_ranges], dtype=np.int64)
    lists = [np.ascontiguousarray(g, dtype=np.int64) for g in ghost_lists]
    off = np.zeros(world + 1, np.int64)
    off[1:] = np.cumsum([len(g) for g in lists])
    allg = np.ascontiguousarray(np.concatenate(lists + [np.empty(0, np.int64)]))
    npeers = C.c_int(0); total = C.c_int64(0)
    L.check(L.lib().pfem_neighbour_plan(world, rank, _p(rs), _p(re), _p(off), _p(allg), C.byref(npeers), C.byref(total),
                                        None, None, None), "pfem_neighbour_plan")
    peers = np.zeros(npeers.value, np.int32); poff = np.zeros(npeers.value + 1, np.int64); gid = np.zeros(total.value, np.int64)
    if npeers.value:
        L.check(L.lib().pfem_neighbour_plan(world, rank, _p(rs), _p(re), _p(off), _p(allg), C.byref(npeers), C.byref(total),
                                            _p(peers), _p(poff), _p(gid)), "pfem_neighbour_plan")
    return peers, poff, gid


def write_temp_dat(path, val, ii=None, ind=None):
    """The drivers' ``temp.dat`` dump (tetrapoissonparallelimpl1.F:935-942; value-only form :1031-1046 of the
    elasticity driver when ``ii``/``ind`` are None)."""
    val = _f64(val)
    if ii is not None:
        ii = np.ascontiguousarray(ii, dtype=np.int64); ind = np.ascontiguousarray(ind, dtype=np.int64)
    L.check(L.lib().pfem_write_temp_dat(str(path).encode(), val.size, _p(ii), _p(ind), _p(val)), "pfem_write_temp_dat")


def writeoutputvtk(ndim, coords, elemNodeConn, elem_procid, soln, fileName, ndof=None):
    """MODULE WriterVTK: writeoutputvtk (writervtk.F:33).  ``coords (ndim,nNode)``, ``elemNodeConn
    (npElem,nElem)`` 0-based, ``soln (nNode, ndof)`` or flat by node id."""
    coords = _f64(coords); conn = _i32(elemNodeConn); pid = _i32(elem_procid)
    soln = _f64(soln)
    nNode, nElem = coords.shape[1], conn.shape[1]
    if ndof is None:
        ndof = soln.size // nNode
    L.check(L.lib().pfem_write_vtk(str(fileName).encode(), ndim, nElem, nNode, conn.shape[0], ndof, _p(coords), _p(conn),
                                   _p(pid), _p(soln.ravel())), "writeoutputvtk")
