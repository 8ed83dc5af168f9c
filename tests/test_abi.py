"""The drop-in boundary: libpfem_amd.so loads, exports every symbol include/pfem_amd.h declares,
and refuses device work loudly when there is no GPU (no CPU fallback exists)."""
import ctypes as C
import os
import re

import pytest

import pfemfort_amd as pf
from pfemfort_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("pfem_amd.h", "pfem_amd_diag.h")):
    names = []
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:int|char)\s*\*?\s*(pfem_\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_boundary_header_holds_the_boundary_only():
    """include/pfem_amd.h = what a maintainer of the reference binds (INTEGRATION.md maps every entry point to the interface it
    replaces); introspection, measurement and lab knobs live in include/pfem_amd_diag.h."""
    boundary, diag = declared_symbols(("pfem_amd.h",)), declared_symbols(("pfem_amd_diag.h",))
    assert not set(boundary) & set(diag)
    for word in ("_bench", "_profile", "_selftest", "_info", "_layout", "_aggregates", "get_spmv", "get_timings", "eval_elems"):
        assert not [n for n in boundary if word in n and n not in ("pfem_device_info", "pfem_solver_print_info")], word
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in boundary if n not in integration]
    assert not missing, f"boundary entry points INTEGRATION.md does not map to a reference interface: {missing}"


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 45
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pfem_amd*.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)


def test_version_and_strerror():
    L = _lib.lib()
    assert L.pfem_version() == 100
    assert b"Negative Jacobian" in L.pfem_strerror(3)
    assert b"no CPU path" in L.pfem_strerror(5)


def test_library_does_not_link_the_oracle_or_torch():
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "torch" not in out and "amdhip64" in out


@pytest.mark.skipif(pf.device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(pf.PfemError) as ei:
        pf.PetscSolver().initialise(10, 10)
    assert ei.value.code == _lib.ERR_NOGPU and "no CPU fallback" in str(ei.value)
    with pytest.raises(pf.PfemError):
        pf.device_info(0)
    with pytest.raises(pf.PfemError) as ei:
        pf.device_memory(0)
    assert ei.value.code == _lib.ERR_NOGPU


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pfemfort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pfem_oracle" not in src and "liboracle" not in src and "orc_" not in src, f


@pytest.mark.skipif(pf.device_count() > 0, reason="a GPU is present")
def test_rccl_binding_loads_and_fails_loudly_without_gpu():
    """librccl is bound at run time (dlopen, no link dependency): on a box without a GPU the library is found, its
    symbols resolve, and ncclGetUniqueId's failure comes back as PFEM_ERR_COMM with RCCL's own text -- no crash."""
    from pfemfort_amd.solver import rccl_unique_id
    with pytest.raises(pf.PfemError) as ei:
        rccl_unique_id()
    assert ei.value.code == 9 and "ncclGetUniqueId" in str(ei.value)
