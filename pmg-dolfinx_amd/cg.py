"""Host mirror of ``acc::CGSolver`` (``src/cg.hpp:93-250``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .vector import Layout, Vector


class CGSolver:
    def __init__(self, layout: Layout, bs: int = 1):  # :99-105
        self.layout = layout
        h = vp()
        call("pmg_cg_create", C.byref(h), layout.handle)
        self._handle = h
        self._max_iter = 0

    @property
    def handle(self):
        return self._handle

    def set_max_iterations(self, max_iter: int):  # :107-113
        self._max_iter = int(max_iter)
        call("pmg_cg_set_max_iterations", self._handle, int(max_iter))

    def set_tolerance(self, tolerance: float):  # :114
        call("pmg_cg_set_tolerance", self._handle, float(tolerance))

    def store_coefficients(self, val: bool):  # :116
        call("pmg_cg_store_coefficients", self._handle, int(bool(val)))

    def set_flexible(self, val: bool):
        """Polak-Ribiere beta for a preconditioner that is not a fixed linear operator (the V-cycle
        with a Krylov coarse solver); not in the reference."""
        call("pmg_cg_set_flexible", self._handle, int(bool(val)))

    def solve(self, A, x: Vector, b: Vector, verbose: bool = False, preconditioner=None) -> int:  # :147-222
        its = C.c_int()
        call("pmg_cg_solve", self._handle, A.handle, ptr(x.data), ptr(b.data),
             preconditioner.handle if preconditioner is not None else vp(0), C.byref(its), current_stream())
        return its.value

    def _coeffs(self):
        cap = max(self._max_iter, 1)
        a = np.empty(cap)
        b = np.empty(cap)
        n = call("pmg_cg_coefficients", self._handle, a.ctypes.data_as(_lib.c_dp), b.ctypes.data_as(_lib.c_dp), cap)
        return a[:n], b[:n]

    def alphas(self):  # :118
        return self._coeffs()[0]

    def betas(self):  # :119
        return self._coeffs()[1]

    def compute_eigenvalues(self):  # :121-142
        cap = max(self._max_iter, 2)
        e = np.empty(cap)
        n = call("pmg_cg_compute_eigenvalues", self._handle, e.ctypes.data_as(_lib.c_dp), cap)
        return e[:n]

    def residual(self) -> float:  # :144
        out = C.c_double()
        call("pmg_cg_residual", self._handle, C.byref(out))
        return out.value

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_cg_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
