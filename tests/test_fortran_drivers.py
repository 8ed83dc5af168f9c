"""The reference's own driver PROGRAMs compiled in place (`make -C oracle drivers drivers_mpi`) against the
build-owned Fortran modules + libpfem_amd.so: the "drop onto it unchanged" check of the boundary, in the build
container only.  The executables are reference-derived build products (oracle/_ref: never committed, never sent
to the GPU box); what they compute is pinned by the fixtures of tests/golden/drivers (test_golden_drivers.py),
and the Fortran boundary itself runs on the GPU box through the build's own program (test_fortran_boundary.py)."""
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


