"""CPU oracle for the matrix-free p-multigrid hot path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``pmg-dolfinx_amd/``) never falls back to this code.

What it restates (all citations relative to the reference tree,
Wells-Group/pmg-dolfinx @ 2024_08_07):

* the GLL-collocated sum-factorised stiffness operator
  ``src/laplacian.hpp:143-278`` and its geometry tensor ``src/laplacian.hpp:22-113``;
* the assembled-CSR "CPU dolfinx path" the reference compares against
  (``examples/mat_free/main.cpp:270-289``, ``src/csr.hpp:64-110``);
* BLAS-1 semantics of ``src/vector.hpp:333-454``;
* 4th-kind Chebyshev ``src/chebyshev.hpp:46-91``;
* Jacobi-PCG + Lanczos + TQLI ``src/cg.hpp:15-84,121-222``;
* p-prolongation / restriction ``src/interpolate.hpp:21-87,118-178``;
* the V-cycle ``src/pmg.hpp:56-155``;
* the ghost-layer partition ``src/mesh.hpp:16-143`` (see ``oracle/partition`` in
  the host package tests).

Third-party arithmetic the reference delegates to (basix / dolfinx / FFCx,
unpinned ``main`` of Aug 2024, absent from the reference tree) is restated from
its published definitions: (P+1)-point Gauss-Lobatto-Legendre rule on [0,1],
Lagrange basis on the GLL nodes (``gll_warped`` variant = GLL nodes), trilinear
coordinate element, point-evaluation interpolation operator.

Parity pinning: the only executable fixture in the reference for this path is
``python_tests/tqli.py`` (golden eigenvalues, see ``tests/golden``) -- TQLI is
pinned by it; for everything else the reference holds no golden vector or
known-answer test, i.e. PARITY UNPINNED by the reference itself: those parts are
pinned by the analytic known-answer tests listed in ``SURVEY.md`` §8(c)
(7-point stencil at P=1, null space, exact energy of linear fields, symmetry,
mat-free == assembled CSR, polynomial reproduction by prolongation, ...).

Conventions fixed by this build (the reference inherits basix's; all parity
checks are permutation invariant):
  * cell-local dof / quadrature index  t = a*nd^2 + b*nd + c  with ``a`` the x
    index (x slowest), exactly ``thread_id`` of ``src/laplacian.hpp:173``;
  * 1-D nodes in ascending order;
  * global dofs of a box mesh numbered lexicographically, x slowest.
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre as _leg

# --------------------------------------------------------------------------
# 1-D tables (what the reference obtains from basix, src/laplacian.hpp:302-317)
# --------------------------------------------------------------------------


def gll_points_weights(n: int):
    """n-point Gauss-Lobatto-Legendre rule on [0, 1] (n >= 2).

    Restates ``basix::quadrature::make_quadrature(type::gll, interval, ...)``
    (call site ``src/laplacian.hpp:307-309``).  Nodes: +-1 and the roots of
    P'_{n-1}; weights 2 / (n (n-1) P_{n-1}(x)^2); mapped from [-1,1] to [0,1].
    """
    if n < 2:
        raise ValueError("GLL needs at least 2 points")
    m = n - 1
    c = np.zeros(m + 1)
    c[m] = 1.0
    if n == 2:
        x = np.array([-1.0, 1.0])
    else:
        interior = _leg.legroots(_leg.legder(c))
        # Newton polish on P'_m (roots from the companion matrix are ~1e-15 already)
        d1 = _leg.legder(c)
        d2 = _leg.legder(c, 2)
        for _ in range(3):
            interior = interior - _leg.legval(interior, d1) / _leg.legval(interior, d2)
        x = np.concatenate(([-1.0], np.sort(interior), [1.0]))
    # enforce exact symmetry
    x = 0.5 * (x - x[::-1])
    w = 2.0 / (n * (n - 1) * _leg.legval(x, c) ** 2)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_eval_matrix(nodes: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """M[j, k] = l_k(pts[j]) for the Lagrange basis on ``nodes``."""
    nodes = np.asarray(nodes, dtype=np.float64)
    pts = np.asarray(pts, dtype=np.float64)
    n = len(nodes)
    M = np.ones((len(pts), n))
    for k in range(n):
        for m in range(n):
            if m != k:
                M[:, k] *= (pts - nodes[m]) / (nodes[k] - nodes[m])
    return M


def lagrange_deriv_matrix(nodes: np.ndarray) -> np.ndarray:
    """D[q, i] = l_i'(nodes[q]) (row = quadrature point, col = dof).

    This is the second half of the basix ``tabulate(1, ...)`` table used as
    ``dphi`` by the reference (``src/laplacian.hpp:312-317``; indexing
    ``dphi[q * nd + i]`` at ``:198``).
    """
    x = np.asarray(nodes, dtype=np.float64)
    n = len(x)
    # barycentric weights
    bw = np.ones(n)
    for j in range(n):
        for m in range(n):
            if m != j:
                bw[j] /= x[j] - x[m]
    D = np.zeros((n, n))
    for q in range(n):
        for i in range(n):
            if i != q:
                D[q, i] = (bw[i] / bw[q]) / (x[q] - x[i])
        D[q, q] = -np.sum(D[q, :])
    return D


def interpolation_matrix_1d(p_coarse: int, p_fine: int) -> np.ndarray:
    """1-D factor of ``basix::compute_interpolation_operator(Q1, Q2)``
    (call site ``src/interpolate.hpp:118``): M[j, k] = l^coarse_k(x^fine_j)."""
    xc, _ = gll_points_weights(p_coarse + 1)
    xf, _ = gll_points_weights(p_fine + 1)
    return lagrange_eval_matrix(xc, xf)


def interpolation_matrix_3d(p_coarse: int, p_fine: int, tol: float = 1e-12) -> np.ndarray:
    """Dense (N_fine x N_coarse) cell interpolation matrix, entries with
    |v| <= tol dropped as in ``src/interpolate.hpp:119-135``."""
    M1 = interpolation_matrix_1d(p_coarse, p_fine)
    M = np.einsum("ai,bj,ck->abcijk", M1, M1, M1).reshape((p_fine + 1) ** 3, (p_coarse + 1) ** 3)
    M = M.copy()
    M[np.abs(M) <= tol] = 0.0
    return M


# --------------------------------------------------------------------------
# Structured box mesh with tensor-product dof numbering (stands in for dolfinx
# create_box + create_functionspace, examples/pmg/main.cpp:83-87,442-451)
# --------------------------------------------------------------------------


class BoxMesh:
    """Unit-box hex mesh of nx*ny*nz cells (optionally mapped by ``warp``)."""

    def __init__(self, n, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), warp=None):
        if np.isscalar(n):
            n = (int(n),) * 3
        self.n = tuple(int(v) for v in n)
        nx, ny, nz = self.n
        self.ncells = nx * ny * nz
        gx = np.linspace(lo[0], hi[0], nx + 1)
        gy = np.linspace(lo[1], hi[1], ny + 1)
        gz = np.linspace(lo[2], hi[2], nz + 1)
        X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        if warp is not None:
            x = warp(x)
        self.xgeom = np.ascontiguousarray(x, dtype=np.float64)  # [npts, 3]
        # geometry dofmap, tensor-product vertex order k = i*4 + j*2 + l (x slowest)
        cx, cy, cz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        cx, cy, cz = cx.ravel(), cy.ravel(), cz.ravel()
        gd = np.empty((self.ncells, 8), dtype=np.int32)
        for i in range(2):
            for j in range(2):
                for l in range(2):
                    gd[:, i * 4 + j * 2 + l] = ((cx + i) * (ny + 1) + (cy + j)) * (nz + 1) + (cz + l)
        self.geom_dofmap = gd
        self._c = (cx, cy, cz)

    def dof_shape(self, P: int):
        nx, ny, nz = self.n
        return (nx * P + 1, ny * P + 1, nz * P + 1)

    def ndofs(self, P: int) -> int:
        s = self.dof_shape(P)
        return s[0] * s[1] * s[2]

    def dofmap(self, P: int) -> np.ndarray:
        """[ncells, (P+1)^3] int32, local index t = a*nd^2 + b*nd + c."""
        NX, NY, NZ = self.dof_shape(P)
        cx, cy, cz = self._c
        nd = P + 1
        a = np.arange(nd)
        gx = (cx[:, None] * P + a[None, :])  # [ncells, nd]
        gy = (cy[:, None] * P + a[None, :])
        gz = (cz[:, None] * P + a[None, :])
        dm = (gx[:, :, None, None] * NY + gy[:, None, :, None]) * NZ + gz[:, None, None, :]
        return np.ascontiguousarray(dm.reshape(self.ncells, nd**3), dtype=np.int32)

    def boundary_marker(self, P: int) -> np.ndarray:
        """int8 marker, 1 on every dof of an exterior facet
        (``examples/pmg/main.cpp:123-124,173-185``)."""
        NX, NY, NZ = self.dof_shape(P)
        m = np.zeros((NX, NY, NZ), dtype=np.int8)
        m[0, :, :] = m[-1, :, :] = 1
        m[:, 0, :] = m[:, -1, :] = 1
        m[:, :, 0] = m[:, :, -1] = 1
        return m.ravel()

    def dof_coordinates(self, P: int) -> np.ndarray:
        """Physical coordinates of every dof [ndofs, 3] (trilinear map of the
        reference GLL nodes)."""
        nd = P + 1
        xi, _ = gll_points_weights(nd)
        phi = np.stack([1.0 - xi, xi], axis=1)  # [nd, 2]
        N = np.einsum("ai,bj,cl->abcijl", phi, phi, phi).reshape(nd**3, 8)
        xc = self.xgeom[self.geom_dofmap]  # [ncells, 8, 3]
        pts = np.einsum("tk,ckd->ctd", N, xc)
        out = np.zeros((self.ndofs(P), 3))
        out[self.dofmap(P).ravel()] = pts.reshape(-1, 3)
        return out


def geometry_tables(P: int):
    """``dphi_geometry`` [3, nq, 8] and 3-D GLL ``weights`` [nq], nq = (P+1)^3,
    i.e. the arrays ``examples/pmg/main.cpp:216-238`` uploads for the geometry
    kernel.  Trilinear coordinate element, vertex order k = i*4 + j*2 + l."""
    nd = P + 1
    xi, w = gll_points_weights(nd)
    phi = np.stack([1.0 - xi, xi], axis=1)  # [nd, 2]
    dph = np.stack([-np.ones(nd), np.ones(nd)], axis=1)
    dx = np.einsum("ai,bj,cl->abcijl", dph, phi, phi).reshape(nd**3, 8)
    dy = np.einsum("ai,bj,cl->abcijl", phi, dph, phi).reshape(nd**3, 8)
    dz = np.einsum("ai,bj,cl->abcijl", phi, phi, dph).reshape(nd**3, 8)
    w3 = np.einsum("a,b,c->abc", w, w, w).ravel()
    return np.ascontiguousarray(np.stack([dx, dy, dz], axis=0)), np.ascontiguousarray(w3)


def geometry_G(xgeom, geom_dofmap, dphi_geom, weights, cells=None):
    """G[c, q, 0..5] = (K K^T)_{00,01,02,11,12,22} * w_q / detJ with K = adj(J).

    Follows ``src/laplacian.hpp:72-111`` (J at ``:81-87``, K at ``:90-95``, G at
    ``:99-111``) but with the correct cofactor expansion of detJ (the reference's
    ``:97`` is only right for diagonal J -- SURVEY quirk Q1) and indexed by cell
    id (quirk Q2)."""
    if cells is None:
        cells = np.arange(geom_dofmap.shape[0])
    xc = xgeom[geom_dofmap[cells]]  # [nc, 8, 3]
    # J[c,q,i,j] = sum_k x[c,k,i] * dphi[j,q,k]
    J = np.einsum("cki,jqk->cqij", xc, dphi_geom)
    K = np.empty_like(J)
    K[..., 0, 0] = J[..., 1, 1] * J[..., 2, 2] - J[..., 1, 2] * J[..., 2, 1]
    K[..., 0, 1] = -J[..., 0, 1] * J[..., 2, 2] + J[..., 0, 2] * J[..., 2, 1]
    K[..., 0, 2] = J[..., 0, 1] * J[..., 1, 2] - J[..., 0, 2] * J[..., 1, 1]
    K[..., 1, 0] = -J[..., 1, 0] * J[..., 2, 2] + J[..., 1, 2] * J[..., 2, 0]
    K[..., 1, 1] = J[..., 0, 0] * J[..., 2, 2] - J[..., 0, 2] * J[..., 2, 0]
    K[..., 1, 2] = -J[..., 0, 0] * J[..., 1, 2] + J[..., 0, 2] * J[..., 1, 0]
    K[..., 2, 0] = J[..., 1, 0] * J[..., 2, 1] - J[..., 1, 1] * J[..., 2, 0]
    K[..., 2, 1] = -J[..., 0, 0] * J[..., 2, 1] + J[..., 0, 1] * J[..., 2, 0]
    K[..., 2, 2] = J[..., 0, 0] * J[..., 1, 1] - J[..., 0, 1] * J[..., 1, 0]
    detJ = J[..., 0, 0] * K[..., 0, 0] + J[..., 0, 1] * K[..., 1, 0] + J[..., 0, 2] * K[..., 2, 0]
    KKt = np.einsum("cqik,cqjk->cqij", K, K)
    s = weights[None, :] / detJ
    G = np.stack(
        [KKt[..., 0, 0], KKt[..., 0, 1], KKt[..., 0, 2], KKt[..., 1, 1], KKt[..., 1, 2], KKt[..., 2, 2]],
        axis=-1,
    ) * s[..., None]
    return np.ascontiguousarray(G), detJ


# --------------------------------------------------------------------------
# Operator (src/laplacian.hpp)
# --------------------------------------------------------------------------


class Laplacian:
    """CPU restatement of ``acc::MatFreeLaplacian`` (``src/laplacian.hpp:284-526``)
    on a whole (single-rank) dof range.  ``cells`` restricts the cell list."""

    def __init__(self, P, kappa, dofmap, xgeom, geom_dofmap, bc_marker, ndofs=None):
        self.P = int(P)
        self.nd = P + 1
        self.dofmap = np.asarray(dofmap, dtype=np.int64).reshape(-1, self.nd**3)
        self.ncells = self.dofmap.shape[0]
        self.kappa = np.broadcast_to(np.asarray(kappa, dtype=np.float64), (self.ncells,)).copy()
        self.bc = np.asarray(bc_marker).astype(bool)
        self.ndofs = int(ndofs if ndofs is not None else self.bc.shape[0])
        nodes, _ = gll_points_weights(self.nd)
        self.D = lagrange_deriv_matrix(nodes)
        dphi_geom, w3 = geometry_tables(P)
        self.G, self.detJ = geometry_G(np.asarray(xgeom), np.asarray(geom_dofmap), dphi_geom, w3)
        self.w3 = w3
        self._diag = None

    # src/laplacian.hpp:143-278
    def cell_apply(self, ue, cells=None):
        """Element-local y_e = kappa * B^T G B u_e, ue: [nc, nd, nd, nd]."""
        D = self.D
        G = self.G if cells is None else self.G[cells]
        kap = self.kappa if cells is None else self.kappa[cells]
        nd = self.nd
        vx = np.einsum("qi,cijk->cqjk", D, ue)  # :195-199
        vy = np.einsum("qj,cijk->ciqk", D, ue)  # :206-210
        vz = np.einsum("qk,cijk->cijq", D, ue)  # :214-218
        G = G.reshape(-1, nd, nd, nd, 6)
        k = kap[:, None, None, None]
        f0 = k * (G[..., 0] * vx + G[..., 1] * vy + G[..., 2] * vz)  # :233
        f1 = k * (G[..., 1] * vx + G[..., 3] * vy + G[..., 4] * vz)  # :234
        f2 = k * (G[..., 2] * vx + G[..., 4] * vy + G[..., 5] * vz)  # :235
        ye = np.einsum("qi,cqjk->cijk", D, f0)  # :246-251
        ye += np.einsum("qj,ciqk->cijk", D, f1)  # :255-259
        ye += np.einsum("qk,cijq->cijk", D, f2)  # :263-267
        return ye

    def apply(self, x, cells=None):
        """y = A x with the reference's BC semantics: BC columns masked
        (``:186-189``), BC rows ``y[dof] = x[dof]`` (``:273-274``), y zeroed first
        (``:466``)."""
        x = np.asarray(x, dtype=np.float64)
        dm = self.dofmap if cells is None else self.dofmap[cells]
        nd = self.nd
        xm = np.where(self.bc[: x.shape[0]], 0.0, x)
        ue = xm[dm].reshape(-1, nd, nd, nd)
        ye = self.cell_apply(ue, cells)
        y = np.bincount(dm.ravel(), weights=ye.ravel(), minlength=x.shape[0])
        touched = np.zeros(x.shape[0], dtype=bool)
        touched[dm.ravel()] = True
        sel = self.bc[: x.shape[0]] & touched
        y[sel] = x[sel]
        return y

    def element_matrices(self):
        """Dense element matrices [ncells, N, N] (small meshes only)."""
        nd, N = self.nd, self.nd**3
        I = np.eye(N).reshape(N, nd, nd, nd)
        out = np.empty((self.ncells, N, N))
        for c in range(self.ncells):
            ue = I
            ye = self.cell_apply(ue, cells=np.full(N, c))
            out[c] = ye.reshape(N, N).T
        return out

    def assemble_csr(self):
        """The assembled operator with dolfinx BC treatment: BC rows/cols zeroed,
        unit diagonal (``src/csr.hpp:84-86``)."""
        import scipy.sparse as sp

        Ae = self.element_matrices()
        N = self.nd**3
        rows = np.repeat(self.dofmap, N, axis=1).ravel()
        cols = np.tile(self.dofmap, (1, N)).ravel()
        A = sp.coo_matrix((Ae.ravel(), (rows, cols)), shape=(self.ndofs, self.ndofs)).tocsr()
        keep = sp.diags((~self.bc).astype(np.float64))
        A = keep @ A @ keep + sp.diags(self.bc.astype(np.float64))
        return A.tocsr()

    def diagonal(self):
        """Matrix-free diagonal of the BC-treated operator (BC rows = 1).

        For the collocated basis, B_q e_i = (D[qx,a] d(qy,b) d(qz,c), ...), so
        diag_i = sum_q G00 D[q,a]^2 (+ y, z analogues) + 2 (G01 D[a,a] D[b,b] + ...)
        evaluated at q = i for the cross terms."""
        if self._diag is not None:
            return self._diag
        nd = self.nd
        D = self.D
        G = self.G.reshape(-1, nd, nd, nd, 6) * self.kappa[:, None, None, None, None]
        D2 = D * D
        dd = np.diag(D)
        de = np.einsum("qa,cqbk->cabk", D2, G[..., 0])
        de += np.einsum("qb,caqk->cabk", D2, G[..., 3])
        de += np.einsum("qk,cabq->cabk", D2, G[..., 5])
        de += 2.0 * G[..., 1] * dd[None, :, None, None] * dd[None, None, :, None]
        de += 2.0 * G[..., 2] * dd[None, :, None, None] * dd[None, None, None, :]
        de += 2.0 * G[..., 4] * dd[None, None, :, None] * dd[None, None, None, :]
        d = np.bincount(self.dofmap.ravel(), weights=de.ravel(), minlength=self.ndofs)
        d[self.bc] = 1.0
        self._diag = d
        return d

    def diag_inverse(self):
        """What ``get_diag_inverse`` returns (``src/csr.hpp:100-110``)."""
        return 1.0 / self.diagonal()

    def rhs_manufactured(self, dof_coords, k=(2, 3, 4)):
        """GLL-collocated load vector of f = -kappa lap(sin kx pi x sin ky pi y sin kz pi z)
        (``examples/pmg/poisson.py:6-8,30,40``), zero Dirichlet lifting + set_bc
        (``examples/pmg/main.cpp:291-295``)."""
        kx, ky, kz = k
        c = dof_coords
        f = (
            (kx * kx + ky * ky + kz * kz)
            * np.pi**2
            * np.sin(kx * np.pi * c[:, 0])
            * np.sin(ky * np.pi * c[:, 1])
            * np.sin(kz * np.pi * c[:, 2])
        )
        # b_i = sum_cells kappa_c * w_q * detJ_q * f(x_q) at q = i
        wdet = (self.w3[None, :] * self.detJ) * self.kappa[:, None]
        b = np.bincount(self.dofmap.ravel(), weights=(wdet * f[self.dofmap]).ravel(), minlength=self.ndofs)
        b[self.bc] = 0.0
        return b


# --------------------------------------------------------------------------
# BLAS-1 (src/vector.hpp:333-454), single rank: "owned" == everything
# --------------------------------------------------------------------------


def inner_product(a, b):
    return float(np.dot(a, b))


def norm(a):
    return float(np.sqrt(np.dot(a, a)))


# --------------------------------------------------------------------------
# TQLI + CG (src/cg.hpp)
# --------------------------------------------------------------------------


def tqli(d, e):
    """QL-implicit eigenvalues of a symmetric tridiagonal matrix, in place on
    ``d`` (``src/cg.hpp:15-84`` == ``python_tests/tqli.py:7-60``).  Returns 0, or
    -1 after 30 sweeps on one eigenvalue."""
    n = len(d)

    def find_m(l):
        for m in range(l, n - 1):
            dd = abs(d[m]) + abs(d[m + 1])
            if abs(e[m]) + dd == dd:
                return m
        return n - 1

    for l in range(n):
        it = 0
        while True:
            m = find_m(l)
            if m == l:
                break
            if it == 30:
                return -1
            it += 1
            g = (d[l + 1] - d[l]) / (2.0 * e[l])
            r = np.sqrt(g * g + 1.0)
            g = d[m] - d[l] + e[l] / (g + r if g >= 0 else g - r)
            s = c = 1.0
            p = 0.0
            early = False
            for i in range(m - 1, l - 1, -1):
                f = s * e[i]
                b = c * e[i]
                r = np.sqrt(f * f + g * g)
                e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p
                    e[m] = 0.0
                    early = True
                    break
                s = f / r
                c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
            if early:
                continue
            d[l] -= p
            e[l] = g
            e[m] = 0.0
        e[l] = 0.0
    return 0


class CGSolver:
    """Jacobi-PCG exactly as ``src/cg.hpp:147-222`` (note: alpha/beta are stored
    only for iterations that did not hit the tolerance break, ``:206-218``)."""

    def __init__(self):
        self.max_iter = 0
        self.rtol = 0.0
        self.store = False
        self.alphas, self.betas, self.residuals = [], [], []

    def set_max_iterations(self, n):
        self.max_iter = int(n)

    def set_tolerance(self, t):
        self.rtol = float(t)

    def store_coefficients(self, v):
        self.store = bool(v)

    def solve(self, A, x, b, precond=None, flexible=False):
        """``precond`` (optional, not in the reference): callable r -> M^-1 r that
        replaces the hard-wired Jacobi ``pointwise_mult`` of ``:161,192`` (SURVEY.md
        8f-3: the V-cycle from a zero initial guess).  ``flexible``: Polak-Ribiere beta
        ``r_new.(z_new - z_old) / r_old.z_old`` for a preconditioner that is not a fixed
        linear operator."""
        dinv = A.diag_inverse()
        M = (lambda v: v * dinv) if precond is None else precond
        y = A.apply(x)
        r = b - y
        p = M(r)
        zold = p.copy()
        rnorm0 = inner_product(p, r)
        rnorm = rnorm0
        rtol2 = self.rtol * self.rtol
        k = 0
        while k < self.max_iter:
            k += 1
            y = A.apply(p)
            alpha = rnorm / inner_product(p, y)
            x += alpha * p
            r -= alpha * y
            y = M(r)
            rnorm_new = inner_product(r, y)
            beta = rnorm_new / rnorm
            if flexible and precond is not None:
                beta = (rnorm_new - inner_product(r, zold)) / rnorm
                zold = y.copy()
            rnorm = rnorm_new
            if rnorm / rnorm0 < rtol2:
                break
            p = beta * p + y
            if self.store:
                self.alphas.append(alpha)
                self.betas.append(beta)
                self.residuals.append(rnorm)
        return k

    def compute_eigenvalues(self):
        """Lanczos tridiagonal from the CG coefficients + TQLI (``src/cg.hpp:121-142``)."""
        ne = len(self.alphas)
        if ne < 2:
            raise RuntimeError("Insufficient data to compute eigenvalues")
        a, bt = np.array(self.alphas), np.array(self.betas)
        d = 1.0 / a
        e = np.zeros(ne)
        d[1:] += bt[:-1] / a[:-1]
        e[:-1] = np.sqrt(bt[:-1]) / a[:-1]
        if tqli(d, e) == -1:
            raise RuntimeError("Eigenvalue estimate failed")
        return np.sort(d)


def estimate_eig_range(A, ndofs, iters=20, rtol=1e-6):
    """The smoother set-up of ``examples/pmg/main.cpp:306-328``: 20-it Jacobi-CG on
    b = 1, x0 = 0; eig_range = {0.1, 1.1} * lambda_max."""
    cg = CGSolver()
    cg.set_max_iterations(iters)
    cg.set_tolerance(rtol)
    cg.store_coefficients(True)
    x = np.zeros(ndofs)
    b = np.ones(ndofs)
    cg.solve(A, x, b)
    eig = cg.compute_eigenvalues()
    return (0.1 * eig[-1], 1.1 * eig[-1]), eig


# --------------------------------------------------------------------------
# Chebyshev (src/chebyshev.hpp)
# --------------------------------------------------------------------------


class Chebyshev:
    """4th-kind Chebyshev with Jacobi, ``src/chebyshev.hpp:46-91``."""

    def __init__(self, eig_range, max_iter=2):
        self.eig_range = tuple(eig_range)
        self.max_iter = int(max_iter)

    def set_max_iterations(self, n):
        self.max_iter = int(n)

    def solve(self, A, x, b):
        lmax = self.eig_range[1]
        dinv = A.diag_inverse()
        q = A.apply(x)  # :56
        r = b - q  # :57
        z = r * dinv  # :67
        z *= 4.0 / (3.0 * lmax)  # :68
        for i in range(1, self.max_iter + 1):
            x += z  # :73
            q = A.apply(z)  # :76
            r -= q  # :77
            z *= (2.0 * i - 1.0) / (2.0 * i + 3.0)  # :80
            q = r * dinv  # :82
            z += ((8.0 * i + 4.0) / (2.0 * i + 3.0) / lmax) * q  # :83
        self.last_residual = r
        return x


# --------------------------------------------------------------------------
# Transfers (src/interpolate.hpp)
# --------------------------------------------------------------------------


class Interpolator:
    """Cell-wise prolongation / restriction, ``src/interpolate.hpp:93-329``."""

    def __init__(self, p_coarse, p_fine, dofmap_c, dofmap_f, ndofs_c, ndofs_f):
        self.M = interpolation_matrix_3d(p_coarse, p_fine)
        self.dmc = np.asarray(dofmap_c, dtype=np.int64)
        self.dmf = np.asarray(dofmap_f, dtype=np.int64)
        self.nc, self.nf = int(ndofs_c), int(ndofs_f)
        # multiplicity of each fine dof over all cells, :172-178
        self.mult = np.bincount(self.dmf.ravel(), minlength=self.nf).astype(np.float64)

    def interpolate(self, coarse):
        """Prolongation: fine[dofsQ2[j]] = sum_k M[j,k] coarse[dofsQ1[k]] (plain
        store, every sharing cell writes the same value), ``:21-45``."""
        vals = coarse[self.dmc] @ self.M.T  # [ncells, Nf]
        fine = np.zeros(self.nf)
        fine[self.dmf.ravel()] = vals.ravel()
        return fine

    def reverse_interpolate(self, fine):
        """Restriction: coarse[dofsQ1[j]] += sum_k M^T[j,k] fine[d]/mult[d],
        output zeroed first, ``:60-87,270``."""
        w = (fine / np.where(self.mult > 0, self.mult, 1.0))[self.dmf]  # [ncells, Nf]
        vals = w @ self.M  # [ncells, Nc]
        return np.bincount(self.dmc.ravel(), weights=vals.ravel(), minlength=self.nc)


# --------------------------------------------------------------------------
# V-cycle (src/pmg.hpp)
# --------------------------------------------------------------------------


class MultigridPreconditioner:
    """``acc::MultigridPreconditioner::apply`` restated, ``src/pmg.hpp:56-155``.
    Levels are ordered coarse -> fine like the reference's vectors."""

    def __init__(self, operators, solvers, interpolators, bc_marker_coarsest, coarse_solver=None):
        self.A = operators
        self.S = solvers
        self.I = interpolators
        self.bc0 = np.asarray(bc_marker_coarsest).astype(np.float64)
        self.coarse = coarse_solver
        self.rnorm = None

    def apply(self, x, y, compute_rnorm=False):
        L = len(self.A)
        u = [np.zeros(A.ndofs) for A in self.A]  # :63-64
        b = [None] * L
        u[L - 1] = y.copy()  # :65
        b[L - 1] = x.copy()  # :68
        for i in range(L - 1, 0, -1):
            self.S[i].solve(self.A[i], u[i], b[i])  # :83
            r = b[i] - self.A[i].apply(u[i])  # :86-87
            b[i - 1] = self.I[i - 1].reverse_interpolate(r)  # :92
        b[0] = b[0] * (1.0 - self.bc0)  # :100-103
        if self.coarse is not None:
            self.coarse(u[0], b[0])  # :107
        else:
            self.S[0].solve(self.A[0], u[0], b[0])  # :109
        for i in range(L - 1):
            du = self.I[i].interpolate(u[i])  # :123
            u[i + 1] = u[i + 1] + du  # :129
            self.S[i + 1].solve(self.A[i + 1], u[i + 1], b[i + 1])  # :138
        if compute_rnorm:
            r = b[L - 1] - self.A[L - 1].apply(u[L - 1])  # :141-143
            self.rnorm = norm(r)
        return u[L - 1]  # :154


# --------------------------------------------------------------------------
# Convenience: the whole C2-style hierarchy on one box (examples/pmg/main.cpp)
# --------------------------------------------------------------------------


def build_hierarchy(n, orders=(1, 2, 4), kappa=2.0, cheb_its=3, warp=None):
    mesh = BoxMesh(n, warp=warp)
    ops, smoothers, eigs = [], [], []
    for P in orders:
        A = Laplacian(P, kappa, mesh.dofmap(P), mesh.xgeom, mesh.geom_dofmap, mesh.boundary_marker(P))
        ops.append(A)
        rng, _ = estimate_eig_range(A, A.ndofs)
        eigs.append(rng)
        smoothers.append(Chebyshev(rng, cheb_its))
    interps = []
    for l in range(len(orders) - 1):
        interps.append(
            Interpolator(orders[l], orders[l + 1], ops[l].dofmap, ops[l + 1].dofmap, ops[l].ndofs, ops[l + 1].ndofs)
        )
    mg = MultigridPreconditioner(ops, smoothers, interps, mesh.boundary_marker(orders[0]))
    b = ops[-1].rhs_manufactured(mesh.dof_coordinates(orders[-1]))
    return mesh, ops, smoothers, interps, mg, b, eigs
