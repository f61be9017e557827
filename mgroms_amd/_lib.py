"""ctypes binding of libmgx.so (include/mgx.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgx.so")


class MgxError(RuntimeError):
    pass


class Params(C.Structure):
    """namelist /nhparam/ (mg_namelist.f90:37-50)."""
    _fields_ = [("solver_prec", C.c_double), ("solver_maxiter", C.c_int), ("nsmall", C.c_int),
                ("ns_coarsest", C.c_int), ("ns_pre", C.c_int), ("ns_post", C.c_int),
                ("cmatrix", C.c_char * 16), ("relax_method", C.c_char * 16), ("interp_type", C.c_char * 16),
                ("restrict_type", C.c_char * 16), ("aggressive", C.c_int), ("netcdf_output", C.c_int),
                ("bmask", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_int)

_DP = C.POINTER(C.c_double)
_SIGS = {
    "mgx_params_default": (C.c_int, [C.POINTER(Params)]),
    "mgx_read_namelist": (C.c_int, [C.c_char_p, C.POINTER(Params)]),
    "mgx_init": (C.c_int, [C.c_int] * 6 + [C.POINTER(Params)]),
    "mgx_matrices": (C.c_int, [_DP, _DP, _DP, _DP, _DP, C.c_double, C.c_double, C.c_double]),
    "mgx_solve": (C.c_int, [_DP, _DP, _DP, _DP]),
    "mgx_solve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_check_nondivergence": (C.c_int, [_DP, _DP, _DP, _DP]),
    "mgx_clean": (None, []),
    "mgx_solve_p": (C.c_int, [C.c_double, C.c_int, C.POINTER(C.c_int), _DP, _DP]),
    "mgx_fcycle": (C.c_int, []),
    "mgx_vcycle": (C.c_int, [C.c_int]),
    "mgx_vcycle2": (C.c_int, [C.c_int, C.c_int]),
    "mgx_relax": (C.c_int, [C.c_int, C.c_int]),
    "mgx_residual": (C.c_int, [C.c_int, _DP]),
    "mgx_fine2coarse": (C.c_int, [C.c_int]),
    "mgx_coarse2fine": (C.c_int, [C.c_int]),
    "mgx_fill_halo": (C.c_int, [C.c_int, C.c_int]),
    "mgx_compute_rhs": (C.c_int, [_DP, _DP, _DP, _DP]),
    "mgx_testgalerkin": (C.c_int, [C.c_int, _DP, _DP]),
    "mgx_nlevs": (C.c_int, []),
    "mgx_level_dims": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgx_rbseq_window_info": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mgx_rbseq_window_rows": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mgx_level_info": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "mgx_level_table": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_int)]),
    "mgx_get_field": (C.c_int, [C.c_int, C.c_int, _DP]),
    "mgx_set_field": (C.c_int, [C.c_int, C.c_int, _DP]),
    "mgx_set_comm": (C.c_int, [EXCHANGE_FN, ALLREDUCE_FN, ALLGATHER_FN, C.c_void_p]),
    "mgx_set_stream": (C.c_int, [C.c_void_p]),
    "mgx_set_verbose": (C.c_int, [C.c_int]),
    "mgx_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "mgx_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "mgx_synchronize": (C.c_int, []),
    "mgx_print_tictoc": (C.c_int, [C.c_char_p]),
    "mgx_tic": (C.c_int, [C.c_int, C.c_char_p]),
    "mgx_toc": (C.c_int, [C.c_int, C.c_char_p]),
    "mgx_time_relax": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "mgx_selftest_divc": (C.c_int, [_DP, _DP, C.c_int, C.POINTER(C.c_longlong)]),
    "mgx_time_residual": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "mgx_counters": (C.c_int, [C.POINTER(C.c_longlong)]),
    "mgx_p2p_handle_bytes": (C.c_int, []),
    "mgx_p2p_prepare": (C.c_int, [C.c_void_p]),
    "mgx_p2p_connect": (C.c_int, [C.c_void_p, C.c_int]),
    "mgx_p2p_exchanges": (C.c_longlong, []),
    "mgx_p2p_local_pointers": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mgx_p2p_connect_pointers": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]),
    "mgx_instance_create": (C.c_int, []),
    "mgx_instance_select": (C.c_int, [C.c_int]),
    "mgx_instance_current": (C.c_int, []),
    "mgx_instance_destroy": (C.c_int, [C.c_int]),
    "mgx_rccl_unique_id_bytes": (C.c_int, []),
    "mgx_rccl_get_unique_id": (C.c_int, [C.c_void_p]),
    "mgx_rccl_connect": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgx_rccl_disconnect": (C.c_int, []),
    "mgx_rccl_selftest": (C.c_int, []),
    "mgx_transport": (C.c_char_p, []),
    "mgx_last_error": (C.c_char_p, []),
    "mgx_version": (C.c_char_p, []),
}
SYMBOLS = tuple(_SIGS)

_lib = None


def lib():
    """Load libmgx.so (built in-tree by mgroms_amd/csrc/Makefile).  No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgxError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(or `make -C mgroms_amd/csrc`). There is no CPU fallback.")
        # torch first: libmgx.so must bind to the HIP runtime torch has already loaded (one runtime per
        # process; a second copy of libamdhip64 does not see the device)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError = a symbol of include/mgx.h is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MgxError(lib().mgx_last_error().decode())
