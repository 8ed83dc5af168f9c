import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Collection order (the GPU suite is run with -x): the oracle-parity tests first, then the host / ABI
# tests, and every test that starts other processes (ranks sharing the device, mpiexec, bench.py) last,
# so that a problem in the multi-process plumbing cannot keep a parity test from running.
_ORDER = ["test_gpu_parity", "test_gpu_full_size", "test_golden_drivers", "test_oracle", "test_host", "test_abi",
          "test_sanitize", "test_bench_contract", "test_fortran_drivers", "test_fortran_boundary", "test_distributed"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER) - 3
    items.sort(key=rank)          # stable: the order inside a file is kept


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The native pieces must exist; build them if the tree is fresh (never falls back)."""
    import __graft_entry__ as g
    from pfemfort_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        g.build()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
