"""Host mirror of ``acc::Chebyshev`` (``src/chebyshev.hpp:19-106``)."""
from __future__ import annotations

import ctypes as C

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .vector import Layout, Vector


class Chebyshev:
    def __init__(self, layout: Layout, eig_range, bs: int = 1):  # :25-33
        self.layout = layout
        self.eig_range = (float(eig_range[0]), float(eig_range[1]))
        h = vp()
        call("pmg_chebyshev_create", C.byref(h), layout.handle, self.eig_range[0], self.eig_range[1])
        self._handle = h

    @property
    def handle(self):
        return self._handle

    def set_max_iterations(self, max_iter: int):  # :35
        call("pmg_chebyshev_set_max_iterations", self._handle, int(max_iter))

    def solve(self, A, x: Vector, b: Vector, verbose: bool = False):  # :46-91
        call("pmg_chebyshev_solve", self._handle, A.handle, ptr(x.data), ptr(b.data), current_stream())

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_chebyshev_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
