"""Host mirror of the coarse solver slot of the reference (``CoarseSolverType<T>``,
``src/amg.hpp``: PETSc KSPCG + hypre BoomerAMG on the degree-1 level): the library's own
smoothed-aggregation AMG, ``pmg_amg`` of ``include/pmg_amd.h``."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .vector import Vector


class AmgSolver:
    """``solve(x, b)`` of ``src/amg.hpp:67-68`` for the degree-1 operator ``op``.

    Krylov mode (default, the reference's shape): CG on the operator preconditioned by one AMG
    V-cycle, at most ``max_iter`` iterations to ``rtol`` (KSP's defaults 60 / 1e-5).
    ``cycles=n``: stationary mode, n AMG V-cycles from a zero initial guess -- a fixed linear
    operator with no host synchronisation (single rank)."""

    def __init__(self, op, max_iter: int = 60, rtol: float = 1e-5, cycles: int = 0, smoother_iterations: int = 2,
                 global_index=None, n_global=None, distributed_fine_level=None, setup="gathered"):
        """``global_index`` (local -> global dof numbers, owned then ghosts) and ``n_global``: the replicated
        form for several ranks -- the hierarchy built on the gathered global matrix on every rank; by default only
        the levels below the first stay replicated in the solve (``distributed_fine_level``, see
        ``pmg_amg_set_distributed_fine_level``)."""
        self.op = op  # must outlive the handle
        h = vp()
        if global_index is not None:
            gi = np.ascontiguousarray(global_index, dtype=np.int64)
            if gi.size != op.layout.total:
                raise ValueError("global_index must have size_local + num_ghosts entries")
            # setup = "gathered": the global degree-1 matrix on every rank (pmg_amg_create_replicated);
            # "distributed": the first coarsening per rank, only level 1 gathered (pmg_amg_create_distributed)
            call({"gathered": "pmg_amg_create_replicated", "distributed": "pmg_amg_create_distributed"}[setup],
                 C.byref(h), op.handle, gi.ctypes.data_as(C.POINTER(C.c_int64)), int(n_global), current_stream())
        else:
            call("pmg_amg_create", C.byref(h), op.handle, current_stream())
        self._handle = h
        if distributed_fine_level is not None:  # replicated form: level 0 on the partitioned operator (default) or not
            call("pmg_amg_set_distributed_fine_level", h, 1 if distributed_fine_level else 0)
        call("pmg_amg_set_smoother_iterations", h, int(smoother_iterations))
        if cycles > 0:
            call("pmg_amg_set_cycles", h, int(cycles))
        else:
            call("pmg_amg_set_krylov", h, int(max_iter), float(rtol))

    @property
    def handle(self):
        return self._handle

    def solve(self, x: Vector, b: Vector) -> int:
        its = C.c_int()
        call("pmg_amg_solve", self._handle, ptr(x.data), ptr(b.data), C.byref(its), current_stream())
        return its.value

    def cycle(self, x: Vector, b: Vector):
        """x = M b: one V-cycle of the hierarchy from a zero initial guess."""
        call("pmg_amg_cycle", self._handle, ptr(x.data), ptr(b.data), current_stream())

    def num_levels(self) -> int:
        return call("pmg_amg_num_levels", self._handle)

    def level_info(self, level: int):
        rows, nnz, lam = C.c_longlong(), C.c_longlong(), C.c_double()
        call("pmg_amg_level_info", self._handle, int(level), C.byref(rows), C.byref(nnz), C.byref(lam))
        return {"rows": rows.value, "nnz": nnz.value, "lambda_max": lam.value}

    def info(self):
        return [self.level_info(l) for l in range(self.num_levels())]

    def export(self, level: int, which: str = "A"):
        """Host copy of the level's matrix ("A") or of the prolongator from level + 1 ("P") as a
        scipy CSR matrix (tests)."""
        import scipy.sparse as sp

        w = {"A": 0, "P": 1}[which]
        rows, cols, nnz = C.c_longlong(), C.c_longlong(), C.c_longlong()
        call("pmg_amg_export", self._handle, int(level), w, C.byref(rows), C.byref(cols), C.byref(nnz), None, None,
             None)
        rp = np.empty(rows.value + 1, np.int32)
        ci = np.empty(max(nnz.value, 1), np.int32)
        v = np.empty(max(nnz.value, 1), np.float64)
        call("pmg_amg_export", self._handle, int(level), w, None, None, None, rp.ctypes.data_as(_lib.c_ip),
             ci.ctypes.data_as(_lib.c_ip), v.ctypes.data_as(_lib.c_dp))
        return sp.csr_matrix((v[: nnz.value], ci[: nnz.value], rp), shape=(rows.value, cols.value))

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_amg_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
