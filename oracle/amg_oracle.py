"""CPU restatement of the solve phase of the library's algebraic multigrid (``csrc/amg.hip``).

TEST INFRASTRUCTURE ONLY.  The reference's coarse solver is PETSc KSPCG + hypre BoomerAMG
(``src/amg.hpp:33-47``) -- third-party arithmetic with no fixture, so parity with the reference is
unpinned for this component.  What is pinned here: given the SAME hierarchy (the matrices and
prolongators the library exports, themselves checked against first principles in
``tests/test_gpu_amg.py``: A_0 against the oracle's assembled operator, A_{l+1} = P^T A_l P), the
device cycle must equal this numpy/scipy cycle -- V(k, k) with the 4th-kind Chebyshev / Jacobi
smoother of ``src/chebyshev.hpp:46-91`` (the oracle's own ``Chebyshev``), exact solve on the last
level."""
from __future__ import annotations

import numpy as np


class _CsrOp:
    """What oracle.pmg_oracle.Chebyshev.solve needs of an operator."""

    def __init__(self, A):
        self.A = A
        self._dinv = 1.0 / A.diagonal()

    def apply(self, x):
        return self.A @ x

    def diag_inverse(self):
        return self._dinv


class AmgCycle:
    def __init__(self, As, Ps, lmax, k=2):
        from . import pmg_oracle as po

        self.As, self.Ps, self.k = As, Ps, k
        self.ops = [_CsrOp(A) for A in As]
        self.smoothers = [po.Chebyshev((0.0, lm), k) for lm in lmax]
        self.dense = np.linalg.inv(As[-1].toarray())

    def cycle(self, b, l=0):
        """x = M b, one V(k, k) cycle from a zero initial guess."""
        if l == len(self.As) - 1:
            return self.dense @ b
        A, P = self.As[l], self.Ps[l]
        x = self.smoothers[l].solve(self.ops[l], np.zeros_like(b), b)
        xc = self.cycle(P.T @ (b - A @ x), l + 1)
        x = x + P @ xc
        return self.smoothers[l].solve(self.ops[l], x, b)

    def pcg(self, A_apply, b, rtol, max_iter):
        """CG of src/cg.hpp:147-222 with this cycle as the preconditioner (stops on r.z)."""
        x = np.zeros_like(b)
        r = b - A_apply(x)
        p = self.cycle(r)
        rz0 = rz = p @ r
        its = 0
        while its < max_iter:
            its += 1
            y = A_apply(p)
            alpha = rz / (p @ y)
            x += alpha * p
            r -= alpha * y
            z = self.cycle(r)
            rz_new = r @ z
            beta = rz_new / rz
            rz = rz_new
            if rz / rz0 < rtol * rtol:
                break
            p = beta * p + z
        return x, its
