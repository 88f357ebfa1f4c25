"""Host mirror of ``acc::MultigridPreconditioner`` (``src/pmg.hpp:16-184``)."""
from __future__ import annotations

import ctypes as C

from . import _lib
from ._lib import call, current_stream, ptr, vp
from .laplacian import _dev_i8
from .vector import Vector


class MultigridPreconditioner:
    """p-multigrid V-cycle; levels ordered coarse -> fine (``maps`` of ``:22-24``)."""

    def __init__(self, layouts, bc_marker_coarsest, bs: int = 1):
        self.layouts = list(layouts)
        self.bc0 = _dev_i8(bc_marker_coarsest, self.layouts[0].device)
        arr = (vp * len(self.layouts))(*[l.handle for l in self.layouts])
        h = vp()
        call("pmg_multigrid_create", C.byref(h), len(self.layouts), arr, ptr(self.bc0))
        self._handle = h
        self._keep = {}

    @property
    def handle(self):
        return self._handle

    def _set(self, fn, objs, key):
        objs = list(objs)
        self._keep[key] = objs
        arr = (vp * max(len(objs), 1))(*[o.handle for o in objs])
        call(fn, self._handle, arr)

    def set_solvers(self, solvers):  # :44
        self._set("pmg_multigrid_set_solvers", solvers, "solvers")

    def set_operators(self, operators):  # :48
        self._set("pmg_multigrid_set_operators", operators, "operators")

    def set_interpolators(self, interpolators):  # :50-53
        self._set("pmg_multigrid_set_interpolators", interpolators, "interpolators")

    def set_coarse_solver(self, solver):  # :46
        """``solver``: a :class:`CGSolver` on the coarsest layout (its iteration cap and tolerance
        apply; zero initial guess, like the reference's KSP solve) or ``None`` for the smoother
        (``src/pmg.hpp:106-109``).  The reference's coarse solver is PETSc KSPCG + hypre BoomerAMG
        (``src/amg.hpp``); the AMG preconditioner is third-party and out of scope, the Krylov
        method here is the library's Jacobi-preconditioned CG."""
        from .cg import CGSolver

        if solver is not None and not isinstance(solver, CGSolver):
            raise TypeError("the coarse solver must be a CGSolver (hypre/PETSc AMG is out of scope) or None")
        self._keep["coarse"] = solver
        call("pmg_multigrid_set_coarse_solver", self._handle, solver.handle if solver is not None else None)

    def apply(self, x: Vector, y: Vector, verbose: bool = False):  # :56-155
        """``x`` is the right-hand side, ``y`` the initial guess on entry and the
        result on exit.  Returns the residual norm when ``verbose`` (the
        reference prints it, ``:147-150``), else ``None``."""
        if verbose:
            rn = C.c_double()
            call("pmg_multigrid_apply", self._handle, ptr(x.data), ptr(y.data), C.byref(rn), current_stream())
            return rn.value
        call("pmg_multigrid_apply", self._handle, ptr(x.data), ptr(y.data), None, current_stream())
        return None

    def apply_counts(self):
        n = len(self.layouts)
        arr = (C.c_int * n)()
        call("pmg_multigrid_apply_counts", self._handle, arr, n)
        return list(arr)

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _lib.lib().pmg_multigrid_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
