"""Generates tests/golden/drivers/*.npz from the reference's UNCHANGED driver programs (build container only).

  make -C oracle fixture_drivers            # reference PROGRAMs + reference element/VTK modules, compiled in place,
                                            # over the build-owned solver module and the CPU trace backend
  python tests/golden/make_driver_fixtures.py

For every case the reference driver is run (alone, or under mpiexec with the shim's stand-in partitioner or with
METIS-format partition files) and what IT computed is stored:
  * temp.dat                      -- the driver's own output (tetrapoissonparallelimpl1.F:935-942,
                                     tetraelasticityparallelimpl1.F:1031-1050): free-dof index, OLD node id, value
  * per rank, from the arguments of its calls into the boundary:
      size_local / row_start      -- solverpetsc%initialise (:779)
      insert_idx[n, nsize]        -- MatSetValues(INSERT_VALUES) rows  == ElemDofArray of the rank's elements (:791-802)
      add_idx, vec_idx, vec_val   -- the ADD_VALUES pass: ElemDofArray again and the LIFTED element vectors (:851-880)
  * its / reason of the (oracle) Jacobi-PCG at rtol 1e-12.
Nothing of the reference itself is stored: inputs are data files, outputs are numbers.
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "drivers")
MPIEXEC = "/opt/conda/bin/mpiexec"


def sector_partition(mesh, world, seed, axes=(0, 1)):
    """The irregular METIS-like partition the tests use: elements by angular sector of the centroid (in the plane of
    ``axes``), every node given to a pseudo-random part among those of the elements touching it."""
    cen = mesh.xyz[:, mesh.conn].mean(axis=1)
    a, b = axes
    epid = np.minimum(((np.arctan2(cen[b] + 0.013, cen[a] + 0.007) + np.pi) / (2 * np.pi) * world).astype(np.int32), world - 1)
    touch = np.zeros((world, mesh.nNode), bool)
    for a in range(mesh.conn.shape[0]):
        touch[epid, mesh.conn[a]] = True
    npid = (np.random.default_rng(seed).random((world, mesh.nNode)) * touch).argmax(axis=0).astype(np.int32)
    return epid, npid


def write_mesh(mesh, prefix):
    with open(prefix + "-nodes.dat", "w") as f:
        for i in range(mesh.nNode):
            f.write("%d\t%.8f\t%.8f\t%.8f\n" % (i + 1, *mesh.xyz[:, i]))
    with open(prefix + "-elems.dat", "w") as f:
        for e in range(mesh.nElem):
            f.write("%d\t%d\t%d\t%d\t%d\n" % (e + 1, *(mesh.conn[:, e] + 1)))
    with open(prefix + "-DirichBC.dat", "w") as f:
        for n, d, v in zip(mesh.bc_node, mesh.bc_dof, mesh.bc_val):
            f.write("%d\t%d\t%.8f\n" % (n + 1, d + 1, v))


def read_trace(d, rank, nsize):
    hdr = dict(line.split() for line in open(os.path.join(d, f"pfem_trace.{rank}.txt")))

    def ragged(name):
        a = np.fromfile(os.path.join(d, f"pfem_trace.{rank}.{name}.i32"), np.int32)
        a = a.reshape(-1, nsize + 1)
        assert (a[:, 0] == nsize).all()
        return np.ascontiguousarray(a[:, 1:])

    out = {"size_local": int(hdr["size_local"]), "row_start": int(hdr["row_start"]), "insert_idx": ragged("insert"),
           "add_idx": ragged("add"), "vec_idx": ragged("vecidx"),
           "vec_val": np.fromfile(os.path.join(d, f"pfem_trace.{rank}.vecval.f64")).reshape(-1, nsize)}
    return out, int(hdr["its"]), int(hdr["reason"]), int(hdr["size_global"])


def run_case(name, driver, prefix_src, nranks, nsize, metis=None):
    exe = os.path.join(REF, "fixture_" + driver + ("_mpi" if nranks > 1 else ""))
    with tempfile.TemporaryDirectory() as d:
        for k in ("nodes", "elems", "DirichBC"):
            shutil.copy(f"{prefix_src}-{k}.dat", os.path.join(d, f"m-{k}.dat"))
        env = dict(os.environ, PFEM_KSP_RTOL="1e-12")
        if metis is not None:
            np.savetxt(os.path.join(d, f"m.epart.{nranks}"), metis[0], fmt="%d")
            np.savetxt(os.path.join(d, f"m.npart.{nranks}"), metis[1], fmt="%d")
            env["PFEM_METIS_PREFIX"] = "m"
        cmd = [exe, "m-nodes.dat", "m-elems.dat", "m-DirichBC.dat"]
        if nranks > 1:
            cmd = [MPIEXEC, "-n", str(nranks)] + cmd
        r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "Program is successful" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        temp = np.loadtxt(os.path.join(d, "temp.dat"))
        arrays = {"temp": temp, "nranks": nranks}
        for rank in range(nranks):
            tr, its, reason, size_global = read_trace(d, rank, nsize)
            for k, v in tr.items():
                arrays[f"r{rank}_{k}"] = v
        arrays.update(its=its, reason=reason, size_global=size_global)
        if metis is not None:
            arrays.update(epart=metis[0], npart=metis[1])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"{name}: N={size_global} its={its} ranks={nranks} "
          f"elements per rank {[len(arrays[f'r{r}_insert_idx']) for r in range(nranks)]}")


def main():
    from oracle import pfem_oracle as O
    with tempfile.TemporaryDirectory() as src:
        for k in ("nodes", "elems", "DirichBC"):
            with gzip.open(os.path.join(ROOT, "tests", "golden", "input", f"tet10-{k}.dat.gz"), "rb") as fi, \
                    open(os.path.join(src, f"tet10-{k}.dat"), "wb") as fo:
                shutil.copyfileobj(fi, fo)
        tet10 = O.read_mesh(os.path.join(src, "tet10"))
        for n in (1, 2, 3):
            run_case(f"tet10_poisson_np{n}", "tetrapoissonparallelimpl1", os.path.join(src, "tet10"), n, 4)
        run_case("tet10_poisson_np3_metis", "tetrapoissonparallelimpl1", os.path.join(src, "tet10"), 3, 4,
                 metis=sector_partition(tet10, 3, 5))
        beam = O.gen_box_tets(-0.5, 0.5, 3, 0.0, 6.0, 12, -0.5, 0.5, 3, bc_mode=1, ndof=3)
        write_mesh(beam, os.path.join(src, "beam"))
        for n in (1, 2):
            run_case(f"beam3x12x3_elast_np{n}", "tetraelasticityparallelimpl1", os.path.join(src, "beam"), n, 12)
        run_case("beam3x12x3_elast_np3_metis", "tetraelasticityparallelimpl1", os.path.join(src, "beam"), 3, 12,
                 metis=sector_partition(beam, 3, 7, axes=(0, 2)))


if __name__ == "__main__":
    main()
