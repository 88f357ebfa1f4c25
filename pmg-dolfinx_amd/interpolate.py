"""Host mirror of ``Interpolator<T>`` (``src/interpolate.hpp:93-329``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .laplacian import _dev_i32, node_order_args
from .vector import Layout, Vector


class Interpolator:
    """Prolongation / restriction between degree ``Q1`` (coarse) and ``Q2`` (fine)
    spaces on the same cells.  The reference takes two basix elements
    (``:104-107``); a tensor-product GLL Lagrange element is fully described by its
    degree."""

    def __init__(self, Q1_degree, Q2_degree, Q1_dofmap, Q2_dofmap, l_cells, b_cells, Q1_layout: Layout,
                 Q2_layout: Layout, fine_operator=None, node_order="ascending", perm1d_coarse=None,
                 perm1d_fine=None):
        dev = Q2_layout.device
        self.lc, self.lf = Q1_layout, Q2_layout
        self.dmc = _dev_i32(Q1_dofmap, dev)
        self.dmf = _dev_i32(Q2_dofmap, dev)
        Nc, Nf = (Q1_degree + 1) ** 3, (Q2_degree + 1) ** 3
        ncells = int(self.dmf.numel() // Nf)
        if self.dmc.numel() // Nc != ncells:  # :113-115
            raise ValueError("coarse and fine dofmaps describe different numbers of cells")
        lc = np.ascontiguousarray(l_cells, dtype=np.int32)
        bc = np.ascontiguousarray(b_cells, dtype=np.int32)
        h = vp()
        # with the fine-level operator the transfers share its cell patches (no atomics)
        self._fine_operator = fine_operator
        # cell-local node order of the two dofmaps (the reference's come from basix elements: "basix")
        code, pc = node_order_args(node_order, int(Q1_degree), perm1d_coarse)
        _, pf = node_order_args(node_order, int(Q2_degree), perm1d_fine)
        call("pmg_interpolator_create_ordered", C.byref(h), Q1_layout.handle, Q2_layout.handle,
             int(Q1_degree), int(Q2_degree), ncells, ptr(self.dmc), ptr(self.dmf), lc.ctypes.data_as(_lib.c_ip),
             lc.size, bc.ctypes.data_as(_lib.c_ip), bc.size,
             fine_operator.handle if fine_operator is not None else vp(0), code,
             pc.ctypes.data_as(_lib.c_ip) if pc is not None else None,
             pf.ctypes.data_as(_lib.c_ip) if pf is not None else None, current_stream())
        self._handle = h

    @property
    def handle(self):
        return self._handle

    def interpolate(self, Q1_vector: Vector, Q2_vector: Vector):  # :186-239
        call("pmg_interpolator_interpolate", self._handle, ptr(Q1_vector.data), ptr(Q2_vector.data),
             current_stream())

    def interpolate_add(self, Q1_vector: Vector, Q2_vector: Vector):
        """Q2 += P Q1 in one pass (needs ``fine_operator``)."""
        call("pmg_interpolator_interpolate_add", self._handle, ptr(Q1_vector.data), ptr(Q2_vector.data),
             current_stream())

    def reverse_interpolate(self, Q2_vector: Vector, Q1_vector: Vector):  # :246-303
        call("pmg_interpolator_reverse_interpolate", self._handle, ptr(Q2_vector.data), ptr(Q1_vector.data),
             current_stream())

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_interpolator_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
