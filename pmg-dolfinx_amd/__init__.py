"""MI355X-native matrix-free p-multigrid (host-side mirror of the reference's
operator / solver interface over the C ABI of ``include/pmg_amd.h``).

Import as ``pmg_dolfinx_amd`` (the directory name carries a hyphen; the shim
``pmg_dolfinx_amd.py`` at the repository root loads it under that name).
"""
from . import _lib  # noqa: F401
from .amg import AmgSolver  # noqa: F401
from .cg import CGSolver  # noqa: F401
from .chebyshev import Chebyshev  # noqa: F401
from .interpolate import Interpolator  # noqa: F401
from .laplacian import MatFreeLaplacian, node_permutation, set_merge_threshold  # noqa: F401
from .mesh import (BoxPartition, basix_node_permutation, cell_permutation, default_proc_dims,  # noqa: F401
                   dofmap_in_node_order)
from .pmg import MultigridPreconditioner  # noqa: F401
from .problem import PoissonHierarchy, make_layout  # noqa: F401
from .vector import (Layout, RcclComm, TorchComm, Vector, WindowComm, axpy, copy, inner_product, norm, pointwise_mult, scale,  # noqa: F401
                     squared_norm)
