"""Host mirror of ``acc::MatFreeLaplacian`` (``src/laplacian.hpp:284-526``)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .vector import Layout, Vector


def _dev_i32(a, device):
    import torch

    if hasattr(a, "data_ptr"):
        return a
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def _dev_f64(a, device):
    import torch

    if hasattr(a, "data_ptr"):
        return a
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def _dev_i8(a, device):
    import torch

    if hasattr(a, "data_ptr"):
        return a
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int8)).to(device)


NODE_ORDERS = {"ascending": 0, "endpoints_first": 1, "basix": 1, "custom": 2}


def node_order_args(node_order, degree, perm1d=None):
    """(order code, int32 permutation or None) for the ``*_ordered`` entry points.  ``node_order`` is
    "ascending", "basix" / "endpoints_first" (vertex 0, vertex 1, interior left to right: the order of a
    basix tensor-product element, ``examples/pmg/main.cpp:83-87``) or "custom" with
    ``perm1d[j]`` = ascending position of the caller's 1-D node ``j``."""
    code = NODE_ORDERS[node_order] if isinstance(node_order, str) else int(node_order)
    perm = None
    if code == 2:
        perm = np.ascontiguousarray(perm1d, dtype=np.int32)
        if perm.size != degree + 1:
            raise ValueError("perm1d needs degree + 1 entries")
    return code, perm


def node_permutation(node_order, degree, perm1d=None):
    """perm1d[j] = ascending position of 1-D node j in the given order (``pmg_node_permutation``)."""
    code, perm = node_order_args(node_order, degree, perm1d)
    out = np.zeros(degree + 1, dtype=np.int32)
    call("pmg_node_permutation", code, int(degree), perm.ctypes.data_as(_lib.c_ip) if perm is not None else None,
         out.ctypes.data_as(_lib.c_ip))
    return out


def set_merge_threshold(patch_dofs: int):
    """Launch plan of operators created from now on: interior colours are merged into one atomic
    launch when the interior list has at most ``patch_dofs`` patch dofs (0: always coloured
    launches; negative: the library's measured defaults).  Process-wide; tests and tuning."""
    call("pmg_set_merge_threshold", int(patch_dofs))


class MatFreeLaplacian:
    """y = A x for the GLL-collocated stiffness operator, matrix-free.

    Argument order follows the reference constructor (``:289-297``); arrays may be
    numpy (uploaded here) or device torch tensors (used in place -- they must
    outlive the operator, exactly like the reference's non-owning spans).  The
    coordinate-element tabulation ``dphi_geometry`` and ``G_weights`` of the
    reference are derived from ``degree`` inside the library.
    """

    value_type = np.float64

    def __init__(self, degree, coefficients, dofmap, xgeom, geometry_dofmap, lcells, bcells, bc_marker,
                 layout: Layout, batch_size: int = 0, node_order="ascending", perm1d=None, dphi_geometry=None,
                 G_weights=None):
        if batch_size != 0:
            # src/laplacian.hpp:391-396 recomputes G per batch to save memory; with
            # 288 GB of HBM the tensor is always resident.
            raise ValueError("geometry batching is not supported: G is kept resident in HBM")
        dev = layout.device
        self.layout = layout
        self.degree = int(degree)
        N = (self.degree + 1) ** 3
        self.dofmap = _dev_i32(dofmap, dev)
        self.ncells = int(self.dofmap.numel() // N) if self.degree >= 1 else 0
        self.kappa = _dev_f64(np.broadcast_to(np.asarray(coefficients, dtype=np.float64), (self.ncells,)).copy()
                              if not hasattr(coefficients, "data_ptr") else coefficients, dev)
        self.xgeom = _dev_f64(xgeom, dev)
        self.geom_dofmap = _dev_i32(geometry_dofmap, dev)
        self.bc_marker = _dev_i8(bc_marker, dev)
        if self.bc_marker.numel() != layout.total:
            raise ValueError("bc_marker must have size_local + num_ghosts entries")
        lc = np.ascontiguousarray(lcells, dtype=np.int32)
        bc_ = np.ascontiguousarray(bcells, dtype=np.int32)
        h = vp()
        # cell-local node order of `dofmap` (and of the two tables when given): the reference's arrays come from a
        # basix tensor-product element, i.e. "basix" (src/laplacian.hpp:289-297, examples/pmg/main.cpp:83-87)
        self.node_order, perm = node_order_args(node_order, self.degree, perm1d)
        self.dphi_geometry = _dev_f64(dphi_geometry, dev) if dphi_geometry is not None else None
        self.G_weights = _dev_f64(G_weights, dev) if G_weights is not None else None
        if (self.dphi_geometry is None) != (self.G_weights is None):
            raise ValueError("dphi_geometry and G_weights come together (src/laplacian.hpp:293-294)")
        call("pmg_laplacian_create_ordered", C.byref(h), layout.handle, self.degree, self.ncells, ptr(self.kappa),
             ptr(self.dofmap), ptr(self.xgeom), int(self.xgeom.numel() // 3), ptr(self.geom_dofmap),
             ptr(self.dphi_geometry), ptr(self.G_weights),
             lc.ctypes.data_as(_lib.c_ip), lc.size, bc_.ctypes.data_as(_lib.c_ip), bc_.size, ptr(self.bc_marker),
             self.node_order, perm.ctypes.data_as(_lib.c_ip) if perm is not None else None, current_stream())
        self._handle = h

    @property
    def handle(self):
        return self._handle

    def __call__(self, x: Vector, y: Vector):  # operator()(in, out), :462-482
        call("pmg_laplacian_apply", self._handle, ptr(x.data), ptr(y.data), current_stream())

    def get_diag_inverse(self, diag_inv: Vector):  # :484-488
        call("pmg_laplacian_get_diag_inverse", self._handle, ptr(diag_inv.data), current_stream())

    def set_diag_inverse(self, diag_inv: Vector):  # :490-495
        call("pmg_laplacian_set_diag_inverse", self._handle, ptr(diag_inv.data), current_stream())

    def compute_diag_inverse(self):
        """Matrix-free replacement of the CSR detour of ``examples/pmg/main.cpp:274-279``."""
        call("pmg_laplacian_compute_diag_inverse", self._handle, current_stream())

    def geometry(self):
        """G in the reference layout [ncells, nq, 6] (device tensor)."""
        import torch

        N = (self.degree + 1) ** 3
        out = torch.empty((self.ncells, N, 6), dtype=torch.float64, device=self.layout.device)
        call("pmg_laplacian_get_geometry", self._handle, ptr(out), current_stream())
        return out

    def assemble_rhs(self, f: Vector, b: Vector):
        call("pmg_laplacian_assemble_rhs", self._handle, ptr(f.data), ptr(b.data), current_stream())

    def is_affine(self) -> bool:
        """Every cell is a parallelepiped (constant Jacobian)."""
        return bool(call("pmg_laplacian_is_affine", self._handle))

    def set_geometry_mode(self, mode: str):
        """"stored" (default, the reference's G[cell][q][6] stream) or "affine"
        (one constant tensor per cell; needs ``is_affine()``)."""
        call("pmg_laplacian_set_geometry_mode", self._handle, {"stored": 0, "affine": 1}[mode])

    def launches_per_apply(self) -> int:
        return call("pmg_laplacian_launches_per_apply", self._handle)

    def apply_streams(self) -> int:
        """2 if the interior colour launches run as two halves on two streams (include/pmg_amd.h), else 1."""
        return call("pmg_laplacian_apply_streams", self._handle)

    def chain_available(self) -> bool:
        """Has the operator chains of patches (include/pmg_amd.h, "Chain form")?"""
        return bool(call("pmg_laplacian_chain_available", self._handle))

    def chain_form(self) -> bool:
        return bool(call("pmg_laplacian_chain_form", self._handle))

    def set_chain_form(self, on: bool):
        call("pmg_laplacian_set_chain_form", self._handle, 1 if on else 0)

    def time_kernel(self, x: Vector, y: Vector, reps: int) -> float:
        """Mean milliseconds of one stiffness-kernel launch (one patch colour);
        an operator application issues ``launches_per_apply()`` of them (HIP
        events on the launch stream)."""
        out = C.c_double()
        call("pmg_laplacian_time_kernel", self._handle, ptr(x.data), ptr(y.data), int(reps), C.byref(out),
             current_stream())
        return out.value

    def set_profiling(self, flag: bool):
        """Bracket the stiffness launches of every application with HIP events (in-situ timing)."""
        call("pmg_laplacian_set_profiling", self._handle, 1 if flag else 0)

    def read_profile(self):
        """(summed milliseconds, number of stiffness launches) recorded since the last read."""
        ms, n = C.c_double(), C.c_longlong()
        call("pmg_laplacian_read_profile", self._handle, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_laplacian_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
