"""ctypes binding of libpfem_amd.so (include/pfem_amd.h).

The library is the product: there is no Python/CPU fallback for any device entry point.
Loading fails loudly if the shared object has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C pfemfort_amd/csrc``), and every solver
call raises :class:`PfemError` carrying the C error code and the library's message.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PFEM_AMD_LIB") or os.path.join(_HERE, "libpfem_amd.so")   # override: development A/B builds

# error codes (include/pfem_amd.h)
OK, ERR_ARG, ERR_STATE, ERR_NEG_JAC, ERR_HIP, ERR_NOGPU, ERR_NOMEM, ERR_DIVERGED, ERR_PATTERN, ERR_COMM = range(10)
# element kinds
POISSON_TRIA, POISSON_TET, ELAST_TET, POISSON_TRIA_INLINE, ELAST_TRIA = 1, 2, 3, 4, 5
# solver status (solverpetsc.F:64-68)
SOLVER_EMPTY, PATTERN_OK, INIT_OK, ASSEMBLY_OK, FACTORISE_OK = 1, 2, 3, 4, 5
INSERT_VALUES, ADD_VALUES = 1, 2

NPELEM = {POISSON_TRIA: 3, POISSON_TET: 4, ELAST_TET: 4, POISSON_TRIA_INLINE: 3, ELAST_TRIA: 3}
NDOF = {POISSON_TRIA: 1, POISSON_TET: 1, ELAST_TET: 3, POISSON_TRIA_INLINE: 1, ELAST_TRIA: 2}
NDIM = {POISSON_TRIA: 2, POISSON_TET: 3, ELAST_TET: 3, POISSON_TRIA_INLINE: 2, ELAST_TRIA: 2}


class PfemError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        msg = f"{where}: error {code} ({_strerror(code)})"
        if detail:
            msg += f" -- {detail}"
        super().__init__(msg)


class Timings(C.Structure):
    _fields_ = [("pattern_ms", C.c_double), ("assemble_ms", C.c_double), ("solve_ms", C.c_double),
                ("spmv_ms_total", C.c_double), ("spmv_launches", C.c_int64), ("upload_ms", C.c_double),
                ("event_overhead_ms", C.c_double), ("iface_ms_total", C.c_double), ("scalar_ms_total", C.c_double),
                ("comm_samples", C.c_int64), ("exposed_ms_total", C.c_double),
                ("graph_iterations", C.c_int64), ("host_enqueue_ms", C.c_double), ("host_enqueued_iterations", C.c_int64),
                ("host_comm_ms", C.c_double)]


# host hooks of the communication backend (include/pfem_amd.h, section 5)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64),
                               C.POINTER(C.c_double), C.POINTER(C.c_double))
RCCL_ID_BYTES = 256

_P = C.c_void_p
_I, _L, _D = C.c_int, C.c_int64, C.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors the header 1:1
SIGNATURES = {
    "pfem_version": [],
    "pfem_strerror": [_I],
    "pfem_last_error_string": [],
    "pfem_device_count": [_P],
    "pfem_device_info": [_I, _P, _I, _P, _P, _P],
    "pfem_device_memory": [_I, _P, _P],
    "pfem_poisson_tria_ke": [_P] * 7,
    "pfem_poisson_tet_ke": [_P] * 8,
    "pfem_elast_tet_ke": [_P] * 8,
    "pfem_elast_tria_ke": [_P] * 7,
    "pfem_gen_box_tets": [_D, _D, _I, _D, _D, _I, _D, _D, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "pfem_dof_numbering": [_L, _I, _L, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pfem_renumber_mesh": [_L, _I, _L, _I, _P, _P, _P, _P, _P, _P],
    "pfem_elem_dof_array": [_L, _I, _I, _P, _P, _P],
    "pfem_assy_for_soln": [_L, _I, _P, _P],
    "pfem_partition_box_slabs": [_I, _I, _I, _I, _P, _P],
    "pfem_partition_box_slabs_axis": [_I, _I, _I, _I, _I, _P, _P],
    "pfem_partition_rcb": [_L, _I, _P, _L, _I, _P, _I, _P, _P],
    "pfem_text_table_shape": [C.c_char_p, _L, _P, _P],
    "pfem_text_table_parse": [C.c_char_p, _L, _L, _I, _P],
    "pfem_write_vtk": [C.c_char_p, _I, _L, _L, _I, _I, _P, _P, _P, _P],
    "pfem_write_temp_dat": [C.c_char_p, _L, _P, _P, _P],
    "pfem_solver_create": [_P, _L, _L, _L, _P, _P, _I],
    "pfem_solver_destroy": [_P],
    "pfem_solver_set_stream": [_P, _P],
    "pfem_solver_set_tolerances": [_P, _D, _D, _D, _I],
    "pfem_solver_status": [_P, _P],
    "pfem_solver_set_zero": [_P],
    "pfem_solver_print_info": [_P],
    "pfem_mat_set_values": [_P, _I, _P, _I, _P, _P, _I],
    "pfem_vec_set_values": [_P, _I, _P, _P, _I],
    "pfem_solver_assemble_matrix_and_vector": [_P, _I, _P, _P, _P, _P],
    "pfem_solver_factorise": [_P],
    "pfem_solver_solve": [_P, _P, _P, _P],
    "pfem_solver_factorise_and_solve": [_P, _P, _P, _P],
    "pfem_solver_get_solution": [_P, _P],
    "pfem_solver_get_history": [_P, _P, _I, _P],
    "pfem_mesh_upload": [_P, _I, _L, _P, _L, _P, _P, _P],
    "pfem_box_slab_sizes": [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "pfem_mesh_generate_box": [_P, _I, _D, _D, _I, _D, _D, _I, _D, _D, _I, _I, _I, _I],
    "pfem_box_slab_sizes_axis": [_I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "pfem_mesh_generate_box_axis": [_P, _I, _D, _D, _I, _D, _D, _I, _D, _D, _I, _I, _I, _I, _I],
    "pfem_mesh_download": [_P, _P, _P, _P, _P],
    "pfem_pattern_build": [_P],
    "pfem_assemble": [_P, _P, _P],
    "pfem_solver_set_assembly_mode": [_P, _I],
    "pfem_solver_assembly_info": [_P, _P, _P, _P, _P],
    "pfem_solver_set_spmv_format": [_P, _I],
    "pfem_solver_get_spmv_format": [_P, _P],
    "pfem_solver_get_spmv_row_group": [_P, _P],
    "pfem_solver_get_spmv_gap_table": [_P, _P],
    "pfem_solver_get_spmv_value_dictionary": [_P, _P],
    "pfem_solver_amg_value_dictionaries": [_P, _I, _P, _P],
    "pfem_solver_get_spmv_gap_escapes": [_P, _P],
    "pfem_solver_spmv_bytes": [_P, _P],
    "pfem_solver_set_preconditioner": [_P, _I],
    "pfem_solver_get_preconditioner": [_P, _P],
    "pfem_solver_set_cg_single_reduction": [_P, _I],
    "pfem_solver_amg_info": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pfem_solver_amg_aggregates": [_P, _I, _P],
    "pfem_solver_amg_transfer": [_P, _I, _P, _P, _P, _P, _P, _P],
    "pfem_solver_amg_layout": [_P, _I, _P, _P, _P, _P],
    "pfem_solver_amg_aggregation": [_P, _I, _P, _P],
    "pfem_solver_incidence_patterns": [_P, _P, _P],
    "pfem_solver_amg_comm_counts": [_P, _P, _P],
    "pfem_solver_amg_cycle_profile": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "pfem_solver_amg_pairing": [_P, _P],
    "pfem_solver_set_amg_options": [_P, _I, _I, _D, _D],
    "pfem_solver_set_amg_cycle": [_P, _I],
    "pfem_solver_amg_cycle": [_P, _P, _P],
    "pfem_eval_elems": [_P, _P, _P, _P, _P],
    "pfem_rhs_add_values": [_P, _L, _P, _P],
    "pfem_matrix_info": [_P, _P, _P, _P, _P],
    "pfem_get_local_to_global": [_P, _P],
    "pfem_get_csr": [_P, _P, _P, _P],
    "pfem_get_rhs": [_P, _P],
    "pfem_spmv": [_P, _P, _P],
    "pfem_bench_spmv": [_P, _I, _P],
    "pfem_get_timings": [_P, _P],
    "pfem_solver_profile_spmv": [_P, _I],
    "pfem_neighbour_plan": [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "pfem_solver_set_neighbours": [_P, _I, _P, _P, _P],
    "pfem_rccl_unique_id": [_P],
    "pfem_solver_set_comm_rccl": [_P, _I, _I, _P],
    "pfem_solver_set_comm_host": [_P, _I, _I, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN, _P],
    "pfem_solver_comm_bench": [_P, C.c_int64, _I, _P, _P],
    "pfem_solver_comm_bench_sizes": [_P, C.c_int64, C.c_int64, _I, _I, _P, _P],
    "pfem_solver_set_comm_peer": [_P, _I, _I, HOST_ALLREDUCE_FN, HOST_EXCHANGE_FN, _P],
    "pfem_solver_comm_shutdown": [_P],
    "pfem_solver_comm_info": [_P, _P, _P, _P, _P],
    "pfem_solver_comm_describe": [_P, _P, _I, _P, _P, _P, _P, _P],
    "pfem_solver_comm_selftest": [_P, _L, _P],
    "pfem_get_ghosts": [_P, _P, _P],
    "pfem_find_ghosts": [_L, _P, _L, _L, _P, _P],
}
_RESTYPES = {"pfem_strerror": C.c_char_p, "pfem_last_error_string": C.c_char_p}

_lib = None


def lib() -> C.CDLL:
    """Load libpfem_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). pfemfort_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = L
    return _lib


def _strerror(code: int) -> str:
    try:
        return lib().pfem_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


def check(rc: int, where: str) -> None:
    if rc != OK:
        raise PfemError(rc, where, lib().pfem_last_error_string().decode())


def device_count() -> int:
    n = C.c_int(0)
    check(lib().pfem_device_count(C.byref(n)), "pfem_device_count")
    return n.value


def device_memory(device: int = 0) -> dict:
    """Free and total device memory in bytes (hipMemGetInfo)."""
    f = C.c_int64(0); t = C.c_int64(0)
    check(lib().pfem_device_memory(device, C.byref(f), C.byref(t)), "pfem_device_memory")
    return {"free_bytes": f.value, "total_bytes": t.value}


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    cu = C.c_int(0); mem = C.c_int64(0); clk = C.c_int(0)
    check(lib().pfem_device_info(device, name, 256, C.byref(cu), C.byref(mem), C.byref(clk)), "pfem_device_info")
    return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value, "clock_khz": clk.value}
