"""The oracle against everything that can pin it without dolfinx: the reference's
one executable fixture (TQLI golden vectors generated from
python_tests/tqli.py) and the analytic known-answer tests of SURVEY.md 8(c).
Both restatements (numpy, C/OpenMP) are checked, and against each other."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import pmg_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


def warp(x):
    return x + 0.03 * np.sin(3.0 * x[:, [1, 2, 0]])


def twist(x):
    """Genuinely trilinear cells (J varies inside a cell; detJ needs the full expansion)."""
    y = x.copy()
    y[:, 0] += 0.12 * x[:, 1] * x[:, 2]
    y[:, 1] += 0.10 * x[:, 0] * x[:, 2] + 0.05 * x[:, 0] * x[:, 1] * x[:, 2]
    y[:, 2] += 0.08 * x[:, 0] * x[:, 1]
    return y


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "tqli_golden.json")) as f:
        return json.load(f)


def test_tqli_golden(golden, built):
    """python_tests/tqli.py:63-99: reference routine's output and scipy's answer."""
    for case in golden["cases"]:
        d, e = np.array(case["d"]), np.array(case["e"])
        dn = d.copy()
        assert po.tqli(dn, e.copy()) == 0
        assert np.allclose(dn, case["tqli_d_out"], rtol=1e-13, atol=0)  # same algorithm, same order
        assert np.allclose(np.sort(dn), case["sorted_eigs_scipy"])  # the reference's own assertion
        rc, dc = co.tqli(d, e)
        assert rc == 0 and np.allclose(dc, case["tqli_d_out"], rtol=1e-13, atol=0)
    assert np.allclose(np.sort(golden["cases"][0]["tqli_d_out"])[[0, -1]], [0.02144096, 1.90275902], atol=1e-8)


def test_gll_rule():
    for n in range(2, 10):
        x, w = po.gll_points_weights(n)
        assert x[0] == 0.0 and x[-1] == 1.0 and abs(w.sum() - 1) < 1e-15
        for deg in range(2 * n - 2):  # exact to degree 2n-3
            assert abs(w @ x**deg - 1.0 / (deg + 1)) < 1e-14
        D = po.lagrange_deriv_matrix(x)
        assert np.abs(D @ np.ones(n)).max() < 1e-12
        assert np.abs(D @ x ** (n - 1) - (n - 1) * x ** (n - 2)).max() < 1e-11


def test_p1_is_seven_point_stencil(built):
    """KAT 1: P=1 with the 2-point GLL (trapezoid) rule is kappa*h*(6u - sum nb)."""
    n, kap = 8, 2.0
    m = po.BoxMesh(n)
    nobc = np.zeros(m.ndofs(1), dtype=np.int8)
    U = np.random.default_rng(0).standard_normal((n + 1,) * 3)
    st = kap / n * (6 * U[1:-1, 1:-1, 1:-1] - U[:-2, 1:-1, 1:-1] - U[2:, 1:-1, 1:-1] - U[1:-1, :-2, 1:-1]
                    - U[1:-1, 2:, 1:-1] - U[1:-1, 1:-1, :-2] - U[1:-1, 1:-1, 2:])
    for A in (po.Laplacian(1, kap, m.dofmap(1), m.xgeom, m.geom_dofmap, nobc),
              co.CLevel(1, kap, m.dofmap(1), m.xgeom, m.geom_dofmap, nobc)):
        y = A.apply(U.ravel()).reshape(U.shape)
        assert np.abs(y[1:-1, 1:-1, 1:-1] - st).max() < 1e-13
        assert np.abs(A.apply(np.ones(U.size))).max() < 1e-13  # KAT 2: null space


@pytest.mark.parametrize("wf", [warp, twist])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 6, 8])
def test_matfree_equals_assembled_and_c(P, wf, built):
    """KATs 4, 5, 6, 10 and numpy == C: symmetry, mat-free vs own CSR (mirror of
    --mat_comp, examples/mat_free/main.cpp:270-289), diagonal, BC semantics."""
    n = 2 if P > 4 else 3
    m = po.BoxMesh(n, warp=wf)
    bc = m.boundary_marker(P)
    A = po.Laplacian(P, 2.0, m.dofmap(P), m.xgeom, m.geom_dofmap, bc)
    if wf is twist:  # the volume of the twisted box from the quadrature of detJ (exact: detJ is a polynomial
        # of degree <= 2 per direction only for P >= 2; here just positivity and variation inside cells)
        assert A.detJ.min() > 0 and np.ptp(A.detJ[0]) > 1e-5 * A.detJ[0].mean()
    Cl = co.CLevel(P, 2.0, m.dofmap(P), m.xgeom, m.geom_dofmap, bc)
    rng = np.random.default_rng(P)
    u, v = rng.standard_normal(A.ndofs), rng.standard_normal(A.ndofs)
    y = A.apply(u)
    assert np.abs(y - Cl.apply(u)).max() < 1e-12 * np.abs(y).max()
    assert np.abs(A.G - Cl.G).max() < 1e-14 * np.abs(A.G).max()
    b = bc.astype(bool)
    assert np.array_equal(y[b], u[b])  # BC rows: y = x
    u2 = u.copy()
    u2[b] += 1.0
    assert np.abs(A.apply(u2)[~b] - y[~b]).max() < 1e-13 * np.abs(y).max()  # BC columns ignored
    if P <= 4:
        M = A.assemble_csr()
        assert np.abs(M @ u - y).max() < 1e-12 * np.abs(y).max()
        assert abs(M - M.T).max() < 1e-12 * abs(M).max()
        assert np.abs(A.diagonal() - M.diagonal()).max() < 1e-12 * M.diagonal().max()
    uu, vv = u * ~b, v * ~b
    assert abs(vv @ A.apply(uu) - uu @ A.apply(vv)) < 1e-11 * np.linalg.norm(A.apply(uu)) * np.linalg.norm(vv)
    assert np.abs(A.diagonal() - Cl.diagonal()).max() < 1e-12 * A.diagonal().max()


def test_linear_field_energy():
    """KAT 3: u = x on the unit cube: u^T A u = kappa, interior rows vanish."""
    for P in (1, 2, 3, 5):
        m = po.BoxMesh(3)
        A = po.Laplacian(P, 2.0, m.dofmap(P), m.xgeom, m.geom_dofmap, np.zeros(m.ndofs(P), dtype=np.int8))
        xc = m.dof_coordinates(P)[:, 0]
        y = A.apply(xc)
        assert abs(xc @ y - 2.0) < 1e-12
        interior = ~m.boundary_marker(P).astype(bool)
        assert np.abs(y[interior]).max() < 1e-12


def test_polynomial_energy():
    """u = x^P lies in the space and (u')^2 has degree 2P-2 <= 2P-1, which the (P+1)-point GLL rule
    integrates exactly: u^T A u = kappa * int_0^1 (P x^(P-1))^2 dx = kappa P^2 / (2P - 1), for the numpy
    and the C restatement (uniform box, several cells per direction)."""
    for P in (1, 2, 3, 4, 6, 8):
        m = po.BoxMesh((3, 2, 2) if P < 6 else (2, 1, 1))
        nobc = np.zeros(m.ndofs(P), dtype=np.int8)
        u = m.dof_coordinates(P)[:, 0] ** P
        for A in (po.Laplacian(P, 2.0, m.dofmap(P), m.xgeom, m.geom_dofmap, nobc),
                  co.CLevel(P, 2.0, m.dofmap(P), m.xgeom, m.geom_dofmap, nobc)):
            assert abs(u @ A.apply(u) - 2.0 * P * P / (2 * P - 1)) < 1e-11 * P * P


def _independent_element_matrix(P, xv, kappa):
    """K_ij = kappa sum_q w_q |det J_q| (J_q^-T grad phi_i(q)) . (J_q^-T grad phi_j(q)) on ONE trilinear cell
    with vertices xv [8, 3] (k = i*4 + j*2 + l), built without anything the oracle uses: GLL nodes from
    numpy's Legendre class, 1-D Lagrange polynomials as explicit np.poly1d products, the full 3-D gradient
    table [nq, N, 3] (no D x I x I factorisation, no sum factorisation), numpy.linalg for J^-1 and det.
    The formula is src/laplacian.hpp:72-111 (G = J^-1 J^-T det J w) composed with :195-270 (B^T G B)."""
    from numpy.polynomial import legendre as L

    n = P + 1
    # GLL nodes on [-1, 1]: +-1 and the roots of P'_{n-1}; weights 2 / (n (n-1) P_{n-1}(x)^2)
    c = np.zeros(n)
    c[-1] = 1.0
    xi = np.concatenate([[-1.0], np.sort(L.legroots(L.legder(c))), [1.0]])
    w = 2.0 / (n * (n - 1) * L.legval(xi, c) ** 2)
    xi, w = 0.5 * (xi + 1.0), 0.5 * w  # -> [0, 1]
    lag = []
    for a in range(n):
        others = np.delete(xi, a)
        lag.append(np.poly1d(others, r=True) / np.prod(xi[a] - others))
    val = np.array([[lag[a](x) for a in range(n)] for x in xi])           # val[q, a] = l_a(x_q)
    der = np.array([[lag[a].deriv()(x) for a in range(n)] for x in xi])   # der[q, a] = l_a'(x_q)
    N = n**3
    Kmat = np.zeros((N, N))
    for qa in range(n):
        for qb in range(n):
            for qc in range(n):
                x, y, z = xi[qa], xi[qb], xi[qc]
                # trilinear coordinate map: J[i][j] = d x_i / d X_j
                J = np.zeros((3, 3))
                for i in range(2):
                    for j in range(2):
                        for l in range(2):
                            px, py, pz = (x if i else 1 - x), (y if j else 1 - y), (z if l else 1 - z)
                            dx, dy, dz = (1.0 if i else -1.0), (1.0 if j else -1.0), (1.0 if l else -1.0)
                            g = np.array([dx * py * pz, px * dy * pz, px * py * dz])
                            J += np.outer(xv[i * 4 + j * 2 + l], g)
                Jinv = np.linalg.inv(J)
                grad = np.empty((N, 3))  # reference gradient of every basis function at this point
                for a in range(n):
                    for b in range(n):
                        for cc in range(n):
                            grad[(a * n + b) * n + cc] = (der[qa, a] * val[qb, b] * val[qc, cc],
                                                          val[qa, a] * der[qb, b] * val[qc, cc],
                                                          val[qa, a] * val[qb, b] * der[qc, cc])
                pg = grad @ Jinv  # rows: J^-T grad
                Kmat += kappa * w[qa] * w[qb] * w[qc] * abs(np.linalg.det(J)) * (pg @ pg.T)
    return Kmat


@pytest.mark.parametrize("P", [1, 2, 3, 4])
def test_element_matrix_independent(P, built):
    """The sum-factorised element contraction of the oracle (numpy and C) against an element matrix
    assembled from first principles on a genuinely trilinear (twisted) cell."""
    m = po.BoxMesh(1, warp=lambda x: twist(x) + 0.07 * x[:, [2, 0, 1]] ** 2)
    nobc = np.zeros(m.ndofs(P), dtype=np.int8)
    Kref = _independent_element_matrix(P, m.xgeom[m.geom_dofmap[0]], 1.7)
    A = po.Laplacian(P, 1.7, m.dofmap(P), m.xgeom, m.geom_dofmap, nobc)
    Ke = A.element_matrices()[0]
    assert np.abs(Ke - Kref).max() < 1e-12 * np.abs(Kref).max()
    Cl = co.CLevel(P, 1.7, m.dofmap(P), m.xgeom, m.geom_dofmap, nobc)
    Kc = np.stack([Cl.apply(e) for e in np.eye(m.ndofs(P))], axis=1)  # one cell: dof = local index
    assert np.array_equal(m.dofmap(P)[0], np.arange(m.ndofs(P)))
    assert np.abs(Kc - Kref).max() < 1e-12 * np.abs(Kref).max()
    # and the assembled operator of a two-cell mesh is the sum of two such matrices
    m2 = po.BoxMesh((2, 1, 1), warp=twist)
    A2 = po.Laplacian(P, 1.0, m2.dofmap(P), m2.xgeom, m2.geom_dofmap, np.zeros(m2.ndofs(P), dtype=np.int8))
    M = np.zeros((m2.ndofs(P),) * 2)
    for c in range(2):
        dm = m2.dofmap(P)[c]
        M[np.ix_(dm, dm)] += _independent_element_matrix(P, m2.xgeom[m2.geom_dofmap[c]], 1.0)
    u = np.random.default_rng(P).standard_normal(m2.ndofs(P))
    assert np.abs(A2.apply(u) - M @ u).max() < 1e-12 * np.abs(M @ u).max()


def test_transfers():
    """KAT 7: prolongation reproduces coarse polynomials; restriction = P^T
    (mirror of python_tests/interpolation_matrix.py:65,78)."""
    m = po.BoxMesh(2, warp=None)
    for pc, pf in ((1, 2), (2, 4), (1, 3), (3, 6)):
        I = po.Interpolator(pc, pf, m.dofmap(pc), m.dofmap(pf), m.ndofs(pc), m.ndofs(pf))
        cc, cf = m.dof_coordinates(pc), m.dof_coordinates(pf)
        poly = lambda c: c[:, 0] ** pc + 2 * c[:, 1] ** pc * c[:, 2] - c[:, 2] ** (pc - 1)  # noqa: E731
        assert np.abs(I.interpolate(poly(cc)) - poly(cf)).max() < 1e-13
        rng = np.random.default_rng(pc)
        a, b = rng.standard_normal(I.nc), rng.standard_normal(I.nf)
        assert abs(b @ I.interpolate(a) - I.reverse_interpolate(b) @ a) < 1e-11
        # C restatement
        lc = co.CLevel(pc, 2.0, m.dofmap(pc), m.xgeom, m.geom_dofmap, m.boundary_marker(pc))
        lf = co.CLevel(pf, 2.0, m.dofmap(pf), m.xgeom, m.geom_dofmap, m.boundary_marker(pf))
        ci = co.CInterp(lc, lf)
        assert np.abs(ci.interpolate(a) - I.interpolate(a)).max() < 1e-13
        assert np.abs(ci.reverse_interpolate(b) - I.reverse_interpolate(b)).max() < 1e-12


def test_lanczos_estimate_p1():
    """KAT 8: lambda_max(D^-1 A) = 1 + cos(pi h) for the 7-point stencil."""
    m = po.BoxMesh(16)
    A = po.Laplacian(1, 2.0, m.dofmap(1), m.xgeom, m.geom_dofmap, m.boundary_marker(1))
    (lo, hi), eig = po.estimate_eig_range(A, A.ndofs)
    lam = 1 + np.cos(np.pi / 16)
    assert eig[-1] <= lam * (1 + 1e-12) and eig[-1] > 0.95 * lam and hi == 1.1 * eig[-1]


def test_chebyshev_contracts():
    m = po.BoxMesh(4, warp=warp)
    A = po.Laplacian(2, 2.0, m.dofmap(2), m.xgeom, m.geom_dofmap, m.boundary_marker(2))
    rng, _ = po.estimate_eig_range(A, A.ndofs)
    b = A.rhs_manufactured(m.dof_coordinates(2))
    x = np.zeros(A.ndofs)
    r0 = po.norm(b)
    x = po.Chebyshev(rng, 10).solve(A, x, b)
    assert po.norm(b - A.apply(x)) < 0.5 * r0


def test_vcycle_lean_equals_reference_form_and_converges(built):
    """KAT 9 + the lean V-cycle of the C restatement (the algorithm the GPU runs)
    equals the reference-faithful numpy V-cycle to rounding."""
    n, orders, k = 6, (1, 2, 4), 3
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    cl = [co.CLevel(P, 2.0, mesh.dofmap(P), mesh.xgeom, mesh.geom_dofmap, mesh.boundary_marker(P)) for P in orders]
    ci = [co.CInterp(cl[i], cl[i + 1]) for i in range(len(orders) - 1)]
    cm = co.CMultigrid(cl, ci, [e[1] for e in eigs], k)
    x, xc = np.zeros_like(b), np.zeros_like(b)
    rn = []
    for i in range(4):
        x = mg.apply(b, x, compute_rnorm=True)
        xc = cm.apply(b, xc)
        rn.append(mg.rnorm)
        assert np.abs(x - xc).max() < 1e-12 * np.abs(x).max()
    assert all(rn[i + 1] < 0.3 * rn[i] for i in range(3))  # contraction per cycle
    # discretisation error of the manufactured solution on the unit cube (p=4, h=1/6)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k)
    x = np.zeros_like(b)
    for _ in range(12):
        x = mg.apply(b, x)
    c = mesh.dof_coordinates(4)
    ue = np.sin(2 * np.pi * c[:, 0]) * np.sin(3 * np.pi * c[:, 1]) * np.sin(4 * np.pi * c[:, 2])
    assert np.abs(x - ue).max() < 5e-4


def test_manufactured_solution_order():
    """Error of the collocated discretisation drops ~2^(p+1) per refinement (p=2)."""
    errs = []
    for n in (4, 8):
        m = po.BoxMesh(n)
        A = po.Laplacian(2, 2.0, m.dofmap(2), m.xgeom, m.geom_dofmap, m.boundary_marker(2))
        b = A.rhs_manufactured(m.dof_coordinates(2), k=(1, 1, 1))
        cg = po.CGSolver()
        cg.set_max_iterations(400)
        cg.set_tolerance(1e-12)
        x = np.zeros(A.ndofs)
        cg.solve(A, x, b)
        c = m.dof_coordinates(2)
        ue = np.sin(np.pi * c[:, 0]) * np.sin(np.pi * c[:, 1]) * np.sin(np.pi * c[:, 2])
        errs.append(np.abs(x - ue).max())
    assert errs[0] / errs[1] > 6.0
