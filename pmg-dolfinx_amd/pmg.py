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
        """The reference's CoarseSolver concept is "anything with ``solve(x, b)``" (``src/amg.hpp:67``,
        called at ``src/pmg.hpp:106-107``); ``None`` restores the level-0 smoother (``:108-109``).
        Wired natively: :class:`AmgSolver` (the library's AMG, the role of hypre BoomerAMG + KSPCG in
        the reference) and :class:`CGSolver` (Jacobi-preconditioned, its iteration cap and tolerance
        apply, zero initial guess).  Any other object with ``solve(Vector, Vector)`` is called back
        from inside the cycle on two vectors of the coarsest layout."""
        from .amg import AmgSolver
        from .cg import CGSolver

        self._keep["coarse"] = solver
        if solver is None:
            call("pmg_multigrid_set_coarse_callback", self._handle, _lib.COARSE_FN(), None)
            call("pmg_multigrid_set_coarse_solver", self._handle, None)
            call("pmg_multigrid_set_coarse_amg", self._handle, None)
        elif isinstance(solver, AmgSolver):
            call("pmg_multigrid_set_coarse_amg", self._handle, solver.handle)
        elif isinstance(solver, CGSolver):
            call("pmg_multigrid_set_coarse_solver", self._handle, solver.handle)
        elif hasattr(solver, "solve"):
            cx, cb = Vector(self.layouts[0]), Vector(self.layouts[0])
            lay = self.layouts[0].handle

            def bridge(user, x, b, stream):
                try:  # owned entries in and out with the library's own copy kernel, on the cycle's stream
                    call("pmg_vec_copy", lay, ptr(cb.data), vp(b), vp(stream))
                    call("pmg_vec_copy", lay, ptr(cx.data), vp(x), vp(stream))
                    solver.solve(cx, cb)
                    call("pmg_vec_copy", lay, vp(x), ptr(cx.data), vp(stream))
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    import sys
                    import traceback

                    traceback.print_exc(file=sys.stderr)
                    return 1

            self._keep["coarse_cb"] = _lib.COARSE_FN(bridge)
            self._keep["coarse_vecs"] = (cx, cb)
            call("pmg_multigrid_set_coarse_callback", self._handle, self._keep["coarse_cb"], None)
        else:
            raise TypeError("the coarse solver needs a solve(x, b) method")

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

    def set_graph(self, enable=True):
        """Replay the cycle as a hipGraph (captured on first use per (rhs, y) pair) where nothing in it
        needs the host: ``True`` / ``False``, or ``None`` for the library's default (eager on one rank; replayed
        on several ranks when the capture holds nothing but kernels); see ``pmg_multigrid_set_graph``."""
        call("pmg_multigrid_set_graph", self._handle, -1 if enable is None else (1 if enable else 0))

    def graph_replays(self) -> int:
        return call("pmg_multigrid_graph_replays", self._handle)

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
