"""ctypes binding of ``lib/libpmg_amd.so`` (C ABI: ``include/pmg_amd.h``).

The library is the product; there is no Python or CPU fallback.  If it has not
been built (``python -c 'import __graft_entry__ as g; g.build()'``) every
attempt to use it raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMG_AMD_LIB") or os.path.join(_HERE, "lib", "libpmg_amd.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_int8)
c_lp = C.POINTER(C.c_int64)
vp = C.c_void_p

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, vp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, c_dp, C.c_int)
COARSE_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, vp)


class PmgError(RuntimeError):
    """A pmg_* call returned non-zero; the message is ``pmg_last_error()``."""


_SIGS = {
    # name: (restype, argtypes)
    "pmg_last_error": (C.c_char_p, []),
    "pmg_version": (C.c_int, []),
    "pmg_gll_table": (C.c_int, [C.c_int, c_dp, c_dp]),
    "pmg_lagrange_derivative_table": (C.c_int, [C.c_int, c_dp]),
    "pmg_interpolation_table": (C.c_int, [C.c_int, C.c_int, c_dp]),
    "pmg_tqli": (C.c_int, [c_dp, c_dp, C.c_int]),
    "pmg_layout_create": (
        C.c_int,
        [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, EXCHANGE_FN, ALLREDUCE_FN, vp],
    ),
    "pmg_layout_destroy": (C.c_int, [vp]),
    "pmg_layout_size_local": (C.c_int32, [vp]),
    "pmg_layout_num_ghosts": (C.c_int32, [vp]),
    "pmg_layout_forward_scatters": (C.c_longlong, [vp]),
    "pmg_layout_set_allreduce_max": (C.c_int, [vp, ALLREDUCE_FN]),
    "pmg_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pmg_comm_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p]),
    "pmg_comm_destroy": (C.c_int, [vp]),
    "pmg_comm_capture_overlaps": (C.c_int, []),
    "pmg_comm_create_windows": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]),
    "pmg_comm_allgather": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "pmg_comm_allreduce_sum": (C.c_int, [vp, vp, C.c_int, vp]),
    "pmg_comm_rank": (C.c_int, [vp]),
    "pmg_comm_size": (C.c_int, [vp]),
    "pmg_layout_set_comm": (C.c_int, [vp, vp, C.c_int32, c_ip, c_ip, c_ip]),
    "pmg_window_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp), C.c_char_p]),
    "pmg_window_fine_grained": (C.c_int, []),
    "pmg_window_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "pmg_window_close": (C.c_int, [vp]),
    "pmg_window_free": (C.c_int, [vp]),
    "pmg_layout_window_describe": (C.c_int, [C.c_int32, c_ip, c_ip, c_lp, c_lp, c_lp]),
    "pmg_layout_set_windows": (
        C.c_int,
        [vp, C.c_int32, c_ip, c_ip, vp, vp, C.POINTER(vp), C.POINTER(vp), c_lp, c_lp, c_lp, c_ip],
    ),
    "pmg_scatter_fwd_begin": (C.c_int, [vp, vp, vp]),
    "pmg_scatter_fwd_end": (C.c_int, [vp, vp, vp]),
    "pmg_scatter_rev_begin": (C.c_int, [vp, vp, vp]),
    "pmg_scatter_rev_end": (C.c_int, [vp, vp, vp]),
    "pmg_vec_set": (C.c_int, [vp, vp, C.c_double, vp]),
    "pmg_vec_scale": (C.c_int, [vp, vp, C.c_double, vp]),
    "pmg_vec_copy": (C.c_int, [vp, vp, vp, vp]),
    "pmg_vec_axpy": (C.c_int, [vp, vp, C.c_double, vp, vp, vp]),
    "pmg_vec_pointwise_mult": (C.c_int, [vp, vp, vp, vp, vp]),
    "pmg_vec_inner_product": (C.c_int, [vp, vp, vp, c_dp, vp]),
    "pmg_vec_squared_norm": (C.c_int, [vp, vp, c_dp, vp]),
    "pmg_vec_norm": (C.c_int, [vp, vp, C.c_int, c_dp, vp]),
    "pmg_laplacian_create": (
        C.c_int,
        [C.POINTER(vp), vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int32, vp, c_ip, C.c_int32, c_ip, C.c_int32, vp, vp],
    ),
    "pmg_laplacian_create_with_tables": (
        C.c_int,
        [C.POINTER(vp), vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, c_ip, C.c_int32, c_ip,
         C.c_int32, vp, vp],
    ),
    "pmg_laplacian_create_ordered": (
        C.c_int,
        [C.POINTER(vp), vp, C.c_int, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, c_ip, C.c_int32, c_ip,
         C.c_int32, vp, C.c_int, c_ip, vp],
    ),
    "pmg_laplacian_node_order": (C.c_int, [vp]),
    "pmg_node_permutation": (C.c_int, [C.c_int, C.c_int, c_ip, c_ip]),
    "pmg_gll_table_ordered": (C.c_int, [C.c_int, C.c_int, c_ip, c_dp, c_dp]),
    "pmg_lagrange_derivative_table_ordered": (C.c_int, [C.c_int, C.c_int, c_ip, c_dp]),
    "pmg_interpolation_table_ordered": (C.c_int, [C.c_int, C.c_int, C.c_int, c_ip, c_ip, c_dp]),
    "pmg_laplacian_destroy": (C.c_int, [vp]),
    "pmg_laplacian_apply": (C.c_int, [vp, vp, vp, vp]),
    "pmg_laplacian_get_diag_inverse": (C.c_int, [vp, vp, vp]),
    "pmg_laplacian_set_diag_inverse": (C.c_int, [vp, vp, vp]),
    "pmg_laplacian_compute_diag_inverse": (C.c_int, [vp, vp]),
    "pmg_laplacian_get_geometry": (C.c_int, [vp, vp, vp]),
    "pmg_laplacian_assemble_rhs": (C.c_int, [vp, vp, vp, vp]),
    "pmg_laplacian_degree": (C.c_int, [vp]),
    "pmg_laplacian_is_affine": (C.c_int, [vp]),
    "pmg_laplacian_set_geometry_mode": (C.c_int, [vp, C.c_int]),
    "pmg_laplacian_launches_per_apply": (C.c_int, [vp]),
    "pmg_laplacian_apply_streams": (C.c_int, [vp]),
    "pmg_laplacian_chain_available": (C.c_int, [vp]),
    "pmg_laplacian_chain_form": (C.c_int, [vp]),
    "pmg_laplacian_set_chain_form": (C.c_int, [vp, C.c_int]),
    "pmg_set_merge_threshold": (C.c_int, [C.c_longlong]),
    "pmg_laplacian_set_profiling": (C.c_int, [vp, C.c_int]),
    "pmg_laplacian_read_profile": (C.c_int, [vp, c_dp, C.POINTER(C.c_longlong)]),
    "pmg_laplacian_time_kernel": (C.c_int, [vp, vp, vp, C.c_int, c_dp, vp]),
    "pmg_chebyshev_create": (C.c_int, [C.POINTER(vp), vp, C.c_double, C.c_double]),
    "pmg_chebyshev_destroy": (C.c_int, [vp]),
    "pmg_chebyshev_set_max_iterations": (C.c_int, [vp, C.c_int]),
    "pmg_chebyshev_solve": (C.c_int, [vp, vp, vp, vp, vp]),
    "pmg_cg_create": (C.c_int, [C.POINTER(vp), vp]),
    "pmg_cg_destroy": (C.c_int, [vp]),
    "pmg_cg_set_max_iterations": (C.c_int, [vp, C.c_int]),
    "pmg_cg_set_tolerance": (C.c_int, [vp, C.c_double]),
    "pmg_cg_store_coefficients": (C.c_int, [vp, C.c_int]),
    "pmg_cg_set_flexible": (C.c_int, [vp, C.c_int]),
    "pmg_cg_solve": (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(C.c_int), vp]),
    "pmg_cg_coefficients": (C.c_int, [vp, c_dp, c_dp, C.c_int]),
    "pmg_cg_compute_eigenvalues": (C.c_int, [vp, c_dp, C.c_int]),
    "pmg_cg_residual": (C.c_int, [vp, c_dp]),
    "pmg_interpolator_create": (
        C.c_int,
        [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int32, vp, vp, c_ip, C.c_int32, c_ip, C.c_int32, vp],
    ),
    "pmg_interpolator_create_with_operator": (
        C.c_int,
        [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int32, vp, vp, c_ip, C.c_int32, c_ip, C.c_int32, vp, vp],
    ),
    "pmg_interpolator_create_ordered": (
        C.c_int,
        [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int32, vp, vp, c_ip, C.c_int32, c_ip, C.c_int32, vp, C.c_int,
         c_ip, c_ip, vp],
    ),
    "pmg_interpolator_interpolate_add": (C.c_int, [vp, vp, vp, vp]),
    "pmg_interpolator_destroy": (C.c_int, [vp]),
    "pmg_interpolator_interpolate": (C.c_int, [vp, vp, vp, vp]),
    "pmg_interpolator_reverse_interpolate": (C.c_int, [vp, vp, vp, vp]),
    "pmg_multigrid_create": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp), vp]),
    "pmg_multigrid_destroy": (C.c_int, [vp]),
    "pmg_multigrid_set_operators": (C.c_int, [vp, C.POINTER(vp)]),
    "pmg_multigrid_set_solvers": (C.c_int, [vp, C.POINTER(vp)]),
    "pmg_multigrid_set_interpolators": (C.c_int, [vp, C.POINTER(vp)]),
    "pmg_multigrid_set_coarse_solver": (C.c_int, [vp, vp]),
    "pmg_multigrid_set_coarse_callback": (C.c_int, [vp, COARSE_FN, vp]),
    "pmg_multigrid_set_coarse_amg": (C.c_int, [vp, vp]),
    "pmg_amg_create": (C.c_int, [C.POINTER(vp), vp, vp]),
    "pmg_amg_create_replicated": (C.c_int, [C.POINTER(vp), vp, C.POINTER(C.c_int64), C.c_int64, vp]),
    "pmg_amg_create_distributed": (C.c_int, [C.POINTER(vp), vp, C.POINTER(C.c_int64), C.c_int64, vp]),
    "pmg_amg_destroy": (C.c_int, [vp]),
    "pmg_amg_set_smoother_iterations": (C.c_int, [vp, C.c_int]),
    "pmg_amg_set_cycles": (C.c_int, [vp, C.c_int]),
    "pmg_amg_set_krylov": (C.c_int, [vp, C.c_int, C.c_double]),
    "pmg_amg_set_distributed_fine_level": (C.c_int, [vp, C.c_int]),
    "pmg_amg_solve": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int), vp]),
    "pmg_amg_cycle": (C.c_int, [vp, vp, vp, vp]),
    "pmg_amg_num_levels": (C.c_int, [vp]),
    "pmg_amg_level_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), c_dp]),
    "pmg_amg_export": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_longlong), c_ip, c_ip, c_dp]),
    "pmg_laplacian_set_geometry_batch": (C.c_int, [vp, C.c_longlong]),
    "pmg_laplacian_geometry_bytes": (C.c_longlong, [vp]),
    "pmg_multigrid_apply": (C.c_int, [vp, vp, vp, c_dp, vp]),
    "pmg_multigrid_apply_counts": (C.c_int, [vp, C.POINTER(C.c_int), C.c_int]),
    "pmg_multigrid_set_graph": (C.c_int, [vp, C.c_int]),
    "pmg_multigrid_graph_replays": (C.c_longlong, [vp]),
}

# functions whose int return value is a count, not a status
_COUNT_FUNCS = {"pmg_multigrid_graph_replays", "pmg_amg_num_levels", "pmg_laplacian_geometry_bytes", "pmg_comm_rank", "pmg_comm_size", "pmg_comm_capture_overlaps", "pmg_cg_coefficients", "pmg_cg_compute_eigenvalues", "pmg_multigrid_apply_counts", "pmg_version",
                "pmg_laplacian_degree", "pmg_laplacian_launches_per_apply", "pmg_laplacian_apply_streams", "pmg_laplacian_is_affine",
                "pmg_laplacian_chain_available", "pmg_laplacian_chain_form",
                "pmg_laplacian_node_order", "pmg_layout_forward_scatters"}

_lib = None


def exported_symbols():
    """Names ``include/pmg_amd.h`` declares (kept in sync by tests/test_abi.py)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP library has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root. "
                "There is no CPU fallback."
            )
        # torch bundles a HIP runtime with the same soname as the system's (libamdhip64.so.7); the one
        # loaded first serves the whole process.  It has to be torch's: with the system runtime
        # loaded first torch's own libraries no longer find the device.  So torch goes first.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("PMG_AMD_LIB_ALLOW_MISSING") and not hasattr(L, name):
                continue  # tools/time_variants.sh against a library built from an older tree (A/B timing only)
            f = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().pmg_last_error().decode("utf-8", "replace")
        raise PmgError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")


def call(name: str, *args):
    """Call a status-returning entry point and raise ``PmgError`` on failure."""
    rc = getattr(lib(), name)(*args)
    if name in _COUNT_FUNCS:
        if rc < 0:
            check(rc, name)
        return rc
    check(rc, name)
    return rc


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array, as c_void_p."""
    if t is None:
        return vp(0)
    if hasattr(t, "data_ptr"):
        return vp(t.data_ptr())
    return vp(t.ctypes.data)


def current_stream():
    """hipStream_t of torch's current stream on the current device."""
    import torch

    return vp(torch.cuda.current_stream().cuda_stream)
