"""Host side of the product (C ABI functions that need no GPU) against the oracle:
per-element compat routines bit-exact, integer bookkeeping bit-exact."""
import os

import numpy as np
import pytest

import pfemfort_amd as pf
from oracle import pfem_oracle as O
from pfemfort_amd import host as H


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "elements.npz"))


@pytest.fixture(scope="module")
def tet10(golden_dir):
    return H.read_mesh(os.path.join(golden_dir, "input", "tet10"))


def test_per_element_routines_bit_exact(gold, tet10):
    z4, z12 = np.zeros(4), np.zeros(12)
    for e in range(0, 6000, 61):
        nd = tet10.conn[:, e]
        K, F = H.StiffnessResidualPoissonLinearTetra(tet10.xyz[0, nd], tet10.xyz[1, nd], tet10.xyz[2, nd],
                                                     H.POISSON_ELEMDATA, H.TIMEDATA, z4)
        assert np.array_equal(K, gold["tet10_poisson_K"][e]) and np.array_equal(F, gold["tet10_poisson_F"][e])
    xyz, conn = gold["rt_xyz"], gold["rt_conn"]
    for e in range(300):
        nd = conn[:, e]
        K, F = H.StiffnessResidualElasticityLinearTetra(xyz[0, nd], xyz[1, nd], xyz[2, nd], gold["rt_elast_data"],
                                                        H.TIMEDATA, z12)
        assert np.array_equal(K, gold["rt_elast_K"][e]) and np.array_equal(F, gold["rt_elast_F"][e])
        K, F = H.StiffnessResidualPoissonLinearTetra(xyz[0, nd], xyz[1, nd], xyz[2, nd], gold["rt_aniso"], H.TIMEDATA, z4)
        assert np.array_equal(K, gold["rt_poisson_K"][e]) and np.array_equal(F, gold["rt_poisson_F"][e])
    xy, c3 = gold["rtri_xy"], gold["rtri_conn"]
    for e in range(300):
        nd = c3[:, e]
        K, F = H.StiffnessResidualPoissonLinearTria(xy[0, nd], xy[1, nd], gold["rtri_data"], H.TIMEDATA, np.zeros(3))
        assert np.array_equal(K, gold["rtri_K"][e]) and np.array_equal(F, gold["rtri_F"][e])


def test_per_element_valc_residual_matches_oracle(tet10):
    rng = np.random.default_rng(5)
    import ctypes as C
    for e in range(0, 6000, 500):
        nd = tet10.conn[:, e]
        vc = rng.standard_normal(4)
        K, F = H.StiffnessResidualPoissonLinearTetra(tet10.xyz[0, nd], tet10.xyz[1, nd], tet10.xyz[2, nd],
                                                     H.POISSON_ELEMDATA, H.TIMEDATA, vc)
        Ko = np.empty(16); Fo = np.empty(4)
        p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)   # noqa: E731
        x, y, z = (np.ascontiguousarray(tet10.xyz[d, nd]) for d in range(3))
        assert O.lib().orc_poisson_tet_ke(p(x), p(y), p(z), p(O.POISSON_ELEMDATA), p(O.TIMEDATA), p(vc), p(Ko), p(Fo)) == 0
        assert np.array_equal(K.ravel(order="F"), Ko) and np.array_equal(F, Fo)


def test_negative_jacobian_code(tet10):
    nd = tet10.conn[[1, 0, 2, 3], 0]
    with pytest.raises(pf.PfemError) as ei:
        H.StiffnessResidualPoissonLinearTetra(tet10.xyz[0, nd], tet10.xyz[1, nd], tet10.xyz[2, nd],
                                              H.POISSON_ELEMDATA, H.TIMEDATA, np.zeros(4))
    assert ei.value.code == 3


@pytest.mark.parametrize("args", [(-2, 2, 10, -1, 1, 10, -1, 1, 10), (-1, 1, 7, -1, 1, 5, 0, 3, 9)])
def test_generator_equals_oracle(args):
    a, b = H.gen_box_tets(*args), O.gen_box_tets(*args)
    for f in ("xyz", "conn", "bc_node", "bc_dof", "bc_val"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    a, b = H.gen_box_tets(*args, bc_mode=1, ndof=3), O.gen_box_tets(*args, bc_mode=1, ndof=3)
    for f in ("bc_node", "bc_dof", "bc_val"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_generator_slab_is_a_slice_of_the_whole():
    whole = H.gen_box_tets(-1, 1, 4, -1, 1, 5, -1, 3, 8)
    part = H.gen_box_tets(-1, 1, 4, -1, 1, 5, -1, 3, 8, kz=(2, 5))
    per = 6 * 4 * 5
    assert np.array_equal(part.conn, whole.conn[:, 2 * per:5 * per])
    assert np.array_equal(part.xyz, whole.xyz) and np.array_equal(part.bc_val, whole.bc_val)


@pytest.mark.parametrize("ndof", [1, 3])
def test_numbering_equals_oracle(tet10, ndof):
    rng = np.random.default_rng(11)
    bn = np.repeat(tet10.bc_node[::2], ndof).astype(np.int32)
    bd = np.tile(np.arange(ndof, dtype=np.int32), len(bn) // ndof)
    if ndof == 3:
        keep = rng.random(len(bn)) < 0.8           # partially constrained nodes
        bn, bd = bn[keep], bd[keep]
    bv = rng.standard_normal(len(bn))
    for nparts, npid in ((1, None), (4, rng.integers(0, 4, tet10.nNode).astype(np.int32)),
                         (3, H.partition_box_slabs(10, 10, 10, 3)[1])):
        a = H.dof_numbering(tet10.nNode, ndof, bn, bd, bv, nparts, npid)
        b = O.dof_numbering(tet10.nNode, ndof, bn, bd, bv, nparts, npid)
        for f in ("node_map_get_old", "node_map_get_new", "NodeDofArrayNew", "solnApplied", "node_start", "node_end",
                  "row_start", "row_end"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (nparts, f)
        assert a.size_global == b.size_global
        cn = a.node_map_get_new[tet10.conn].astype(np.int32)
        cn2, xn2 = H.renumber_mesh(tet10, a)            # the C gathers vs. plain numpy indexing
        assert np.array_equal(cn2, cn) and np.array_equal(xn2, tet10.xyz[:, a.node_map_get_old])
        assert np.array_equal(H.elem_dof_array(cn, a.NodeDofArrayNew), O.elem_dof_array(cn, b.NodeDofArrayNew))
        assert np.array_equal(H.assy_for_soln(a.NodeDofArrayNew), O.assy_for_soln(b.NodeDofArrayNew))


def test_slab_partition_properties():
    epid, npid = H.partition_box_slabs(4, 3, 10, 4)
    assert epid.min() == 0 and epid.max() == 3 and (np.diff(epid) >= 0).all()
    assert np.bincount(epid).tolist() == [6 * 12 * c for c in (2, 3, 2, 3)]
    m = H.gen_box_tets(0, 1, 4, 0, 1, 3, 0, 1, 10)
    # a node belongs to the lowest part among the elements that touch it
    low = np.full(m.nNode, 99)
    for a in range(4):
        np.minimum.at(low, m.conn[a], epid)
    assert np.array_equal(low, npid)


def test_find_ghosts():
    edof = np.array([[0, 5, 9, -1], [3, 4, 12, 9]], np.int32)
    assert H.find_ghosts(edof, 3, 4).tolist() == [0, 9, 12]
    assert H.find_ghosts(edof, 0, 13).tolist() == []


def test_vtk_writer_byte_identical_to_reference_writer(gold, tet10, golden_dir, tmp_path):
    """Output step: same bytes as the reference's writervtk.F (golden files written by it)."""
    import gzip
    for name, field, ndof in (("tet10_scalar", gold["vtk_scalar"], 1), ("tet10_vector", gold["vtk_vector"], 3)):
        out = tmp_path / f"{name}.vtk"
        H.writeoutputvtk(3, tet10.xyz, tet10.conn, gold["vtk_procid"], field, out, ndof=ndof)
        with gzip.open(os.path.join(golden_dir, f"vtk_{name}.vtk.gz"), "rb") as z:
            assert out.read_bytes() == z.read()


def test_ascii_mesh_reader_equals_numpy(golden_dir):
    """Ingest step: the library's ASCII table reader against numpy.loadtxt on the shipped files."""
    import gzip
    for name in ("tet10-nodes", "tet10-elems", "tet10-DirichBC", "tria20x20-nodes", "tria20x20-elems", "tet100-DirichBC"):
        path = os.path.join(golden_dir, "input", name + ".dat.gz")
        with gzip.open(path, "rt") as f:
            ref = np.loadtxt(f, ndmin=2)
        got = H.read_table(path)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    m = H.read_mesh(os.path.join(golden_dir, "input", "tet10"))
    o = O.read_mesh(os.path.join(golden_dir, "input", "tet10"))
    assert np.array_equal(m.xyz, o.xyz) and np.array_equal(m.conn, o.conn) and np.array_equal(m.bc_val, o.bc_val)


def test_ascii_reader_edge_cases(tmp_path):
    p = tmp_path / "t.dat"
    p.write_text("  1   0.5\t-2.0D+01 \r\n\n 2 1e-3 7  trailing\n3,4,5\n")
    t = H.read_table(str(p))
    assert t.shape == (3, 3) and np.array_equal(t, [[1, 0.5, -20.0], [2, 1e-3, 7], [3, 4, 5]])
    p.write_text("1 2 3\n4 5\n")
    with pytest.raises(pf.PfemError):
        H.read_table(str(p))
    p.write_text("")
    assert H.read_table(str(p)).shape == (0, 0)


def test_elast_tria_host_routine_bit_exact(gold, golden_dir):
    cook = H.read_mesh(os.path.join(golden_dir, "input", "cookmembranetria32"))
    assert cook.force_val.sum() == 100.0 and len(cook.force_node) == 33        # shipped ForceBC file
    for e in range(0, 2048, 11):
        nd = cook.conn[:, e]
        K, F = H.StiffnessResidualElasticityLinearTria(cook.xyz[0, nd], cook.xyz[1, nd], gold["cook_elast_data"],
                                                       H.TIMEDATA, np.zeros(6))
        assert np.array_equal(K, gold["cook_elast_K"][e]) and np.array_equal(F, gold["cook_elast_F"][e])


def test_read_metis_partition_files(tmp_path):
    ep = np.array([0, 2, 1, 1, 0], np.int32); npart = np.array([1, 1, 0, 2], np.int32)
    np.savetxt(tmp_path / "m.epart.3", ep, fmt="%d"); np.savetxt(tmp_path / "m.npart.3", npart, fmt="%d")
    e, n = H.read_metis_partition(str(tmp_path / "m"), 3)
    assert np.array_equal(e, ep) and np.array_equal(n, npart) and e.dtype == np.int32
    np.savetxt(tmp_path / "m.npart.2", npart, fmt="%d"); np.savetxt(tmp_path / "m.epart.2", ep, fmt="%d")
    with pytest.raises(ValueError):
        H.read_metis_partition(str(tmp_path / "m"), 2)          # part id 2 with nParts = 2


@pytest.mark.parametrize("box,bc_mode,ndof", [((5, 4, 6), 0, 1), ((3, 5, 7), 1, 3), ((2, 2, 9), 0, 3)])
def test_box_slab_sizes_match_the_host_bookkeeping(box, bc_mode, ndof):
    """pfem_box_slab_sizes (closed forms) against pfem_partition_box_slabs + pfem_dof_numbering on the whole grid."""
    nEx, nEy, nEz = box
    mesh = H.gen_box_tets(-1, 1, nEx, 0, 2, nEy, -1, 3, nEz, bc_mode=bc_mode, ndof=ndof)
    for nparts in (1, 2, 3):
        epid, npid = H.partition_box_slabs(nEx, nEy, nEz, nparts)
        dm = H.dof_numbering(mesh.nNode, ndof, mesh.bc_node, mesh.bc_dof, mesh.bc_val, nparts, npid)
        assert np.array_equal(dm.node_map_get_old, np.arange(mesh.nNode))          # z-slabs: the renumbering is the identity
        for part in range(nparts):
            sz = H.box_slab_sizes(nEx, nEy, nEz, bc_mode, ndof, nparts, part)
            assert sz["size_global"] == dm.size_global
            assert (sz["row_start"], sz["row_start"] + sz["size_local"]) == (int(dm.row_start[part]), int(dm.row_end[part]))
            assert sz["nElem_local"] == int((epid == part).sum())
            k0, k1 = nEz * part // nparts, nEz * (part + 1) // nparts
            assert sz["nNode_local"] == (nEx + 1) * (nEy + 1) * (k1 - k0 + 1)


@pytest.mark.parametrize("nparts", [1, 2, 3, 5, 8])
def test_rcb_partition_is_balanced_compact_and_deterministic(tet10, nparts, golden_dir):
    """pfem_partition_rcb: parts differ by at most one element per split level, every node's owner assembles an element
    at it, the result does not depend on anything but the mesh, and on the 2-D Cook membrane too."""
    for mesh in (tet10, H.read_mesh(f"{golden_dir}/input/cookmembranetria32")):
        epid, npid = H.partition_rcb(mesh, nparts)
        e2, n2 = H.partition_rcb(mesh, nparts)
        assert np.array_equal(epid, e2) and np.array_equal(npid, n2)
        cnt = np.bincount(epid, minlength=nparts)
        assert cnt.min() >= mesh.nElem // nparts - 4 and cnt.max() <= -(-mesh.nElem // nparts) + 4 and len(cnt) == nparts
        assert npid.min() >= 0 and npid.max() < nparts
        touch = np.zeros((nparts, mesh.nNode), bool)
        for a in range(mesh.conn.shape[0]):
            touch[epid, mesh.conn[a]] = True
        assert touch[npid, np.arange(mesh.nNode)].all()                     # the owner assembles at the node
        # compact: the parts' bounding boxes overlap little -- far fewer interface nodes than a random assignment
        if nparts > 1:
            shared = (touch.sum(axis=0) > 1).sum()
            rnd = np.random.default_rng(0).integers(0, nparts, mesh.nElem)
            t2 = np.zeros((nparts, mesh.nNode), bool)
            for a in range(mesh.conn.shape[0]):
                t2[rnd, mesh.conn[a]] = True
            assert shared < 0.5 * (t2.sum(axis=0) > 1).sum()


def test_ascii_reader_numbers_bit_identical_to_strtod_many_pieces(tmp_path):
    """The reader's fast decimal path (<= 15 significant digits) and its strtod fall-back against Python's float() --
    itself correctly rounded -- on a file large enough to be cut into one piece per thread, with blank lines, ragged
    spacing and records that straddle the cuts."""
    rng = np.random.default_rng(5)
    n = 60000
    toks = []
    for i in range(n):
        k = i % 12
        if k == 0:
            t = "%d" % rng.integers(-2**31, 2**31)
        elif k == 1:
            t = "%.8f" % rng.uniform(-1e3, 1e3)
        elif k == 2:
            t = "%.15g" % rng.uniform(-1, 1)
        elif k == 3:
            t = "%.17g" % rng.uniform(-1, 1)                 # 17 digits: strtod path
        elif k == 4:
            t = "%.6e" % rng.uniform(-1e-30, 1e30)
        elif k == 5:
            t = ("%.6E" % rng.uniform(-1e5, 1e5)).replace("E", "D")
        elif k == 6:
            t = "." + "%d" % rng.integers(0, 10**9)
        elif k == 7:
            t = "%d." % rng.integers(0, 10**9)
        elif k == 8:
            t = "+0.000000000000000000000%d" % rng.integers(1, 10**6)    # > 22 fractional digits
        elif k == 9:
            t = "-0.0"
        elif k == 10:
            t = "123456789012345.6789"                       # 19 digits
        else:
            t = "999999999999999"                            # exactly 15 digits
        toks.append(t)
    lines = []
    for r in range(n // 3):
        sep = ["  ", "\t", ","][r % 3]
        lines.append(" " * (r % 4) + sep.join(toks[3 * r:3 * r + 3]) + (" \r" if r % 5 == 0 else ""))
        if r % 97 == 0:
            lines.append("   ")
    p = tmp_path / "big.dat"
    p.write_text("\n".join(lines) + "\n")
    assert p.stat().st_size > 8 * 65536                      # several pieces
    got = H.read_table(str(p))
    want = np.array([float(t.replace("D", "e")) for t in toks]).reshape(-1, 3)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # no trailing newline, a file of one record, and a token that is not a number
    p.write_text("1 2 3")
    assert np.array_equal(H.read_table(str(p)), [[1, 2, 3]])
    p.write_text("1 2 x3\n")
    with pytest.raises(pf.PfemError):
        H.read_table(str(p))
    p.write_text("1 2 3\n" * 30000 + "1 2\n" + "1 2 3\n" * 30000)      # a short record deep inside another piece
    with pytest.raises(pf.PfemError):
        H.read_table(str(p))


def test_vtk_writer_numbers_identical_to_printf(tmp_path):
    """The writer's short F12.6 path (and its snprintf fall-back near ties / for large values) against Python's
    correctly rounded '%12.6f', over many blocks so that every thread formats some; temp.dat through the same emitter."""
    rng = np.random.default_rng(11)
    n = 50000
    k = rng.integers(-10**7, 10**7, n)
    vals = np.concatenate([
        rng.uniform(-3, 3, n), rng.uniform(-9999.5, 9999.5, n), rng.uniform(-1e-6, 1e-6, n),
        (2 * k + 1) * 0.5e-6,                                # decimal ties (not exact in binary: either side)
        np.array([0.0, -0.0, 0.5e-6, 1.5e-6, 2.5e-6, -0.5e-6, 0.125, 0.0000005, 9998.9999995, 9998.99999949, -9998.9999995,
                  99999.9999994, 123456.789, -123456.789, 1e15, 1e22, -1e-300, 5e-324, 0.4999995, 0.9999995, 9.9999995])])
    nN = len(vals)
    xyz = np.zeros((3, nN)); xyz[0] = vals; xyz[1] = vals[::-1]; xyz[2] = np.roll(vals, 7)
    conn = np.zeros((4, 5), np.int32); conn[:, 3] = [nN - 1, 12345, 0, 7]
    pid = np.array([0, 1, 22, 333, -4], np.int32)
    out = tmp_path / "a.vtk"
    H.writeoutputvtk(3, xyz, conn, pid, vals, str(out))
    lines = out.read_text().split("\n")
    assert lines[4] == "POINTS %10d float" % nN
    for i in list(range(0, nN, 37)) + list(range(nN - 25, nN)):
        assert lines[5 + i] == "%12.6f%12.6f%12.6f" % (xyz[0, i], xyz[1, i], xyz[2, i]), i
    c0 = 5 + nN
    assert lines[c0] == "CELLS %10d%10d" % (5, 25) and lines[c0 + 4] == "%10d%10d%10d%10d%10d" % (4, nN - 1, 12345, 0, 7)
    assert lines[c0 + 6] == "CELL_TYPES%10d" % 5 and lines[c0 + 7] == " 10"
    assert lines[c0 + 15:c0 + 20] == ["%3d" % v for v in pid]
    s0 = lines.index("LOOKUP_TABLE default", c0 + 20) + 1
    assert lines[s0:s0 + nN] == ["%12.6f" % v for v in vals]
    t = tmp_path / "temp.dat"
    H.write_temp_dat(str(t), vals, np.arange(1, nN + 1), np.arange(nN)[::-1] + 1)
    got = t.read_text().split("\n")
    assert got[:-1] == [" %11d %11d   %.16E" % (i + 1, nN - i, v) for i, v in enumerate(vals)] and got[-1] == ""
    H.write_temp_dat(str(t), vals[:9000])
    assert t.read_text() == "".join("   %.16E\n" % v for v in vals[:9000])


def test_library_leaves_the_process_environment_alone():
    """Round 4's static initializer set OMP_WAIT_POLICY / KMP_BLOCKTIME for the whole process (it slowed the oracle's
    libgomp in bench.py's cpu_baseline leg 2.1x); the library now quiets its OWN OpenMP regions through its runtime's API."""
    import ctypes
    import subprocess
    import sys
    code = ("import os, ctypes\n"
            "os.environ.pop('OMP_WAIT_POLICY', None); os.environ.pop('KMP_BLOCKTIME', None)\n"
            "import pfemfort_amd\nfrom pfemfort_amd import host as H\n"
            "H.gen_box_tets(-1, 1, 12, -1, 1, 12, -1, 1, 12)\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "print(libc.getenv(b'OMP_WAIT_POLICY'), libc.getenv(b'KMP_BLOCKTIME'))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_WAIT_POLICY", "KMP_BLOCKTIME")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "None None", (r.stdout, r.stderr[-500:])
