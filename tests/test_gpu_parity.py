"""GPU parity: every entry point of the C ABI against the CPU oracle on the same
seeded inputs.  Tolerances (FP64, atomic-order noise): 1e-12 relative per
operator apply / transfer / vector op, 1e-10 after a Chebyshev solve or a
V-cycle, eigenvalue estimates 1e-8 (the reference's own bar is
1e-9 absolute on norms, test/test_csr.cpp:113)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def warp(x):
    """Curved grid lines, but every cell stays a parallelepiped (each coordinate is a
    sum of 1-D functions): J varies from cell to cell, not inside a cell."""
    return x + 0.03 * np.sin(3.0 * x[:, [1, 2, 0]])


def twist(x):
    """Genuinely trilinear cells: J (and G) vary inside every cell."""
    y = x.copy()
    y[:, 0] += 0.12 * x[:, 1] * x[:, 2]
    y[:, 1] += 0.10 * x[:, 0] * x[:, 2] + 0.05 * x[:, 0] * x[:, 1] * x[:, 2]
    y[:, 2] += 0.08 * x[:, 0] * x[:, 1]
    return y


@pytest.fixture(scope="module")
def pm(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pmg_dolfinx_amd as pm

    torch.cuda.set_device(0)
    return pm


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _single_level(pm, n, P, warped=True, bc=True):
    from oracle import pmg_oracle as po

    wf = {True: warp, False: None, "twist": twist}[warped]
    part = pm.BoxPartition(n, warp=wf)
    lv = part.level(P)
    bcm = lv.bc_marker if bc else np.zeros_like(lv.bc_marker)
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, bcm, layout)
    A = po.Laplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, bcm)
    return part, lv, layout, op, A


def _vec(pm, layout, a):
    v = pm.Vector(layout)
    v.data.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)))
    return v


@pytest.mark.parametrize("mesh", [True, "twist"])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 7, 8])
def test_apply_parity_all_degrees(pm, P, mesh):
    n = (3, 2, 4) if P > 4 else (5, 4, 3)
    part, lv, layout, op, A = _single_level(pm, n, P, warped=mesh)
    assert op.is_affine() == (mesh is True)
    rng = np.random.default_rng(P)
    u = rng.standard_normal(lv.ndofs)
    x, y = _vec(pm, layout, u), pm.Vector(layout)
    y.set(7.0)  # operator() must zero its output (src/laplacian.hpp:466)
    op(x, y)
    ref = A.apply(u)
    assert _relerr(y.data_copy(), ref) < 1e-12
    # geometry tensor and inverse diagonal
    assert _relerr(op.geometry().cpu().numpy(), A.G) < 1e-13
    op.compute_diag_inverse()
    d = pm.Vector(layout)
    op.get_diag_inverse(d)
    assert _relerr(d.data_copy(), A.diag_inverse()) < 1e-12


@pytest.mark.parametrize("P,n", [(1, (8, 8, 16)), (2, (4, 4, 16)), (3, (4, 4, 8)), (4, (4, 4, 8)), (5, (4, 4, 14)),
                                 (6, (2, 4, 8)), (7, (2, 2, 6)), (8, (2, 2, 6)),
                                 # ... and meshes whose patches are all cut short (P = 5: 4 of the 7 cells of an item)
                                 (5, (4, 4, 4)), (6, (2, 4, 4)), (7, (2, 2, 4)), (8, (2, 2, 8))])
def test_apply_parity_full_patches(pm, P, n):
    """Meshes large enough that every patch of the operator is full (at P = 2 a
    wavefront takes 7 cells, which does not divide the 32 cells of a patch), on a
    tensor grid so that the patch builder takes its structured path."""
    part, lv, layout, op, A = _single_level(pm, n, P, warped=False)
    u = np.random.default_rng(100 + P).standard_normal(lv.ndofs)
    x, y = _vec(pm, layout, u), pm.Vector(layout)
    y.set(-3.0)
    op(x, y)
    assert _relerr(y.data_copy(), A.apply(u)) < 1e-12
    op(x, y)  # second application: no dependence on the previous content of y
    assert _relerr(y.data_copy(), A.apply(u)) < 1e-12


def test_config1_seven_point_stencil(pm):
    """BASELINE config 1: 16^3 hexes, P=1 == 7-point finite differences."""
    part, lv, layout, op, A = _single_level(pm, 16, 1, warped=False)
    for u in (np.ones(lv.ndofs), np.random.default_rng(0).standard_normal(lv.ndofs)):
        x, y = _vec(pm, layout, u), pm.Vector(layout)
        op(x, y)
        got = y.data_copy().reshape(17, 17, 17)
        U = u.reshape(17, 17, 17)
        h, kap = 1.0 / 16, 2.0
        # rows whose whole stencil is interior (neighbours not Dirichlet)
        c = U[2:-2, 2:-2, 2:-2]
        st = kap * h * (6 * c - U[1:-3, 2:-2, 2:-2] - U[3:-1, 2:-2, 2:-2] - U[2:-2, 1:-3, 2:-2]
                        - U[2:-2, 3:-1, 2:-2] - U[2:-2, 2:-2, 1:-3] - U[2:-2, 2:-2, 3:-1])
        assert np.abs(got[2:-2, 2:-2, 2:-2] - st).max() < 1e-13 * max(1.0, np.abs(st).max())
        # BC rows: y = x (src/laplacian.hpp:273-274)
        bc = lv.bc_marker.astype(bool)
        assert np.array_equal(y.data_copy()[bc], u[bc])
        assert _relerr(y.data_copy(), A.apply(u)) < 1e-12


def test_blas1(pm):
    part, lv, layout, op, A = _single_level(pm, 3, 3)
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(lv.ndofs), rng.standard_normal(lv.ndofs)
    va, vb, vr = _vec(pm, layout, a), _vec(pm, layout, b), pm.Vector(layout)
    assert abs(pm.inner_product(va, vb) - a @ b) < 1e-12 * np.abs(a * b).sum()
    assert abs(pm.norm(va) - np.linalg.norm(a)) < 1e-13 * np.linalg.norm(a)
    assert abs(pm.squared_norm(va) - a @ a) < 1e-12 * (a @ a)
    assert pm.norm(va, "linf") == np.abs(a).max()
    pm.axpy(vr, -0.75, va, vb)
    assert np.allclose(vr.data_copy(), -0.75 * a + b, rtol=0, atol=1e-15)
    pm.pointwise_mult(vr, va, vb)
    assert np.array_equal(vr.data_copy(), a * b)
    pm.copy(vr, va)
    assert np.array_equal(vr.data_copy(), a)
    pm.scale(vr, 3.0)
    assert np.array_equal(vr.data_copy(), 3.0 * a)
    vr.set(1.5)
    assert np.array_equal(vr.data_copy(), np.full(lv.ndofs, 1.5))
    with pytest.raises(RuntimeError):
        pm.norm(va, "l1")


def test_rhs_and_chebyshev(pm):
    from oracle import pmg_oracle as po

    part, lv, layout, op, A = _single_level(pm, 4, 3)
    op.compute_diag_inverse()
    c = part.dof_coordinates(3)
    assert _relerr(c, po.BoxMesh(4, warp=warp).dof_coordinates(3)) < 1e-14
    fv = 29 * np.pi**2 * np.sin(2 * np.pi * c[:, 0]) * np.sin(3 * np.pi * c[:, 1]) * np.sin(4 * np.pi * c[:, 2])
    f, b = _vec(pm, layout, fv), pm.Vector(layout)
    op.assemble_rhs(f, b)
    bref = A.rhs_manufactured(c)
    assert _relerr(b.data_copy(), bref) < 1e-12
    for k in (1, 2, 3, 5):
        sm = pm.Chebyshev(layout, (0.2, 2.3))
        sm.set_max_iterations(k)
        x0 = np.random.default_rng(k).standard_normal(lv.ndofs)
        x = _vec(pm, layout, x0)
        sm.solve(op, x, b)
        ref = po.Chebyshev((0.2, 2.3), k).solve(A, x0.copy(), bref)
        assert _relerr(x.data_copy(), ref) < 1e-10


def test_cg_and_eigenvalues(pm):
    from oracle import pmg_oracle as po

    part, lv, layout, op, A = _single_level(pm, 16, 1, warped=False)
    op.compute_diag_inverse()
    cg = pm.CGSolver(layout)
    cg.set_max_iterations(20)
    cg.set_tolerance(1e-6)
    cg.store_coefficients(True)
    x, b = pm.Vector(layout), pm.Vector(layout)
    x.set(0.0)
    b.set(1.0)
    its = cg.solve(op, x, b)
    ocg = po.CGSolver()
    ocg.set_max_iterations(20)
    ocg.set_tolerance(1e-6)
    ocg.store_coefficients(True)
    xo = np.zeros(lv.ndofs)
    oits = ocg.solve(A, xo, np.ones(lv.ndofs))
    assert its == oits
    assert np.allclose(cg.alphas(), ocg.alphas, rtol=1e-9)
    assert np.allclose(cg.betas(), ocg.betas, rtol=1e-9)
    assert _relerr(x.data_copy(), xo) < 1e-9
    eig, oeig = cg.compute_eigenvalues(), ocg.compute_eigenvalues()
    assert np.allclose(eig, oeig, rtol=1e-8)
    # closed form for the Jacobi-scaled 7-point stencil: lambda_max = 1 + cos(pi h)
    lam = 1 + np.cos(np.pi / 16)
    assert eig[-1] <= lam * (1 + 1e-12) and eig[-1] > 0.95 * lam
    # too few coefficients -> the reference's runtime_error (src/cg.hpp:125)
    cg2 = pm.CGSolver(layout)
    cg2.set_max_iterations(1)
    with pytest.raises(RuntimeError, match="Insufficient data"):
        cg2.compute_eigenvalues()


@pytest.mark.parametrize("patched", [False, True])
@pytest.mark.parametrize("pc,pf,n,warped", [(1, 2, (3, 2, 2), True), (2, 4, (3, 2, 2), True), (1, 3, (3, 2, 2), True),
                                            (3, 6, (3, 2, 2), True), (4, 8, (3, 2, 2), True),
                                            (2, 4, (4, 4, 16), False), (1, 2, (4, 4, 16), False),
                                            (1, 4, (2, 4, 8), False),
                                            # fine levels whose patches are one item shared by four waves (p = 5, 8)
                                            # or long columns (p = 6), full and cut short
                                            (2, 5, (2, 2, 7), True), (2, 5, (2, 2, 4), False), (3, 7, (2, 2, 6), True),
                                            (3, 6, (2, 2, 8), False), (4, 8, (2, 2, 6), False)])
def test_transfer_parity(pm, pc, pf, n, warped, patched):
    """Both transfer implementations: the cell form (reference-shaped constructor)
    and the patch form that shares the fine operator's patches."""
    from oracle import pmg_oracle as po

    part = pm.BoxPartition(n, warp=warp if warped else None)
    lc, lf = part.level(pc), part.level(pf)
    Lc, Lf = pm.make_layout(lc), pm.make_layout(lf)
    fop = None
    if patched:
        fop = pm.MatFreeLaplacian(pf, 2.0, lf.dofmap, part.xgeom, part.geom_dofmap, lf.lcells, lf.bcells,
                                  lf.bc_marker, Lf)
    ip = pm.Interpolator(pc, pf, lc.dofmap, lf.dofmap, lf.lcells, lf.bcells, Lc, Lf, fine_operator=fop)
    oi = po.Interpolator(pc, pf, lc.dofmap, lf.dofmap, lc.ndofs, lf.ndofs)
    rng = np.random.default_rng(pc * 10 + pf)
    uc, uf = rng.standard_normal(lc.ndofs), rng.standard_normal(lf.ndofs)
    vc, vf = _vec(pm, Lc, uc), pm.Vector(Lf)
    ip.interpolate(vc, vf)
    assert _relerr(vf.data_copy(), oi.interpolate(uc)) < 1e-13
    vf2, vc2 = _vec(pm, Lf, uf), pm.Vector(Lc)
    vc2.set(3.0)  # must be zeroed by reverse_interpolate (src/interpolate.hpp:270)
    ip.reverse_interpolate(vf2, vc2)
    assert _relerr(vc2.data_copy(), oi.reverse_interpolate(uf)) < 1e-12
    # restriction is the transpose of prolongation
    assert abs(uf @ vf.data_copy() - vc2.data_copy() @ uc) < 1e-11 * np.abs(uf).sum()
    if patched:  # fused correction: fine += P coarse (src/pmg.hpp:123-129)
        vf3 = _vec(pm, Lf, uf)
        ip.interpolate_add(vc, vf3)
        assert _relerr(vf3.data_copy(), uf + oi.interpolate(uc)) < 1e-13
        ip.reverse_interpolate(vf2, vc2)  # second call: no dependence on the previous content
        assert _relerr(vc2.data_copy(), oi.reverse_interpolate(uf)) < 1e-12
    else:
        with pytest.raises(RuntimeError):
            ip.interpolate_add(vc, pm.Vector(Lf))


@pytest.mark.parametrize("orders,n", [((1, 2, 4), 4), ((1, 3), 5), ((2, 4), (3, 4, 2)), ((3,), 3),
                                      ((1, 3, 6), 3), ((1, 2, 4, 8), 3),  # (1, 3, 6): BASELINE config 5's levels
                                      ((1, 5), (2, 2, 7)), ((3, 7), (2, 2, 3))])
def test_vcycle_parity(pm, orders, n):
    from oracle import pmg_oracle as po

    k = 3
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, warp=warp)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    for got, ref in zip(h.eig_ranges, eigs):
        assert abs(got[1] - ref[1]) < 1e-8 * ref[1]
    for s, e in zip(sm, h.eig_ranges):
        s.eig_range = e
    assert _relerr(h.rhs[-1].data_copy(), b) < 1e-12
    x = h.new_vector()
    x.set(0.0)
    xo = np.zeros_like(b)
    for cyc in range(3):  # stationary iteration as in examples/pmg/main.cpp:362-367
        rn = h.mg.apply(h.rhs[-1], x, verbose=True)
        xo = mg.apply(b, xo, compute_rnorm=True)
        assert _relerr(x.data_copy(), xo) < 1e-10
        assert abs(rn - mg.rnorm) < 1e-9 * max(mg.rnorm, 1e-30) + 1e-12


def test_config2_full_size_vcycle_against_c_oracle(pm):
    """BASELINE config 2 at its full size -- 64^3 hexes, p = 4 -> 2 -> 1, Chebyshev(3), 17 M fine dofs:
    the V-cycle the bench times (coloured launches on p4 and p2, patch transfers, streaming cache
    policy) against the C/OpenMP oracle, three cycles from x = 0, plus the fine operator alone."""
    from oracle import c_oracle as co

    n, orders, k = 64, (1, 2, 4), 3
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k)
    assert h.operators[-1].launches_per_apply() >= 8  # the coloured path
    part = h.part
    cl = [co.CLevel(p, 2.0, part.level(p).dofmap, part.xgeom, part.geom_dofmap, part.level(p).bc_marker)
          for p in orders]
    ci = [co.CInterp(cl[i], cl[i + 1]) for i in range(len(orders) - 1)]
    cm = co.CMultigrid(cl, ci, [e[1] for e in h.eig_ranges], k)
    u = np.random.default_rng(3).standard_normal(h.levels[-1].ndofs)
    xu, yu = _vec(pm, h.layouts[-1], u), h.new_vector()
    h.operators[-1](xu, yu)
    assert _relerr(yu.data_copy(), cl[-1].apply(u)) < 1e-12
    del xu, yu
    b = h.rhs[-1].data_copy()
    x = h.new_vector()
    x.set(0.0)
    xo = np.zeros_like(b)
    for cyc in range(3):
        h.mg.apply(h.rhs[-1], x)
        cm.apply(b, xo)
        assert _relerr(x.data_copy(), xo) < 1e-10, cyc


def test_vcycle_graph_replay(pm):
    """pmg_multigrid_set_graph: the cycle replayed as a hipGraph gives the eager cycle's iterates (same
    kernels in the same order), also as the preconditioner of CG and with the stationary AMG coarse
    solver; configurations that need the host inside the cycle fall back to eager launches."""
    h = pm.PoissonHierarchy(8, (1, 2, 4), kappa=2.0, cheb_its=3, warp=warp)
    b = h.rhs[-1]

    def cycles(k):
        x = h.new_vector()
        x.set(0.0)
        for _ in range(k):
            rn = h.mg.apply(b, x, verbose=True)
        return x.data_copy(), rn

    for coarse in (None, "amg"):
        amg = pm.AmgSolver(h.operators[0], cycles=1) if coarse else None
        h.mg.set_coarse_solver(amg)
        h.mg.set_graph(False)
        eager, rn_e = cycles(3)
        h.mg.set_graph(True)
        n0 = h.mg.graph_replays()
        graph, rn_g = cycles(3)
        assert h.mg.graph_replays() == n0 + 3
        assert _relerr(graph, eager) < 1e-13 and abs(rn_g - rn_e) < 1e-12 * rn_e
        # a changed smoother degree is a different graph, not a stale replay
        h.smoothers[-1].set_max_iterations(2)
        h.mg.set_graph(False)
        eager2, _ = cycles(2)
        h.mg.set_graph(True)
        graph2, _ = cycles(2)
        assert _relerr(graph2, eager2) < 1e-13 and _relerr(graph2, eager[: graph2.size]) > 1e-6
        h.smoothers[-1].set_max_iterations(3)
        # as the preconditioner of CG
        sols = []
        for flag in (False, True):
            h.mg.set_graph(flag)
            cg = pm.CGSolver(h.layouts[-1])
            cg.set_max_iterations(30)
            cg.set_tolerance(1e-10)
            xs = h.new_vector()
            xs.set(0.0)
            sols.append((cg.solve(h.operators[-1], xs, b, preconditioner=h.mg), xs.data_copy()))
        assert sols[0][0] == sols[1][0] and _relerr(sols[1][1], sols[0][1]) < 1e-11
    # a Krylov coarse solver synchronises the host: eager, no replay
    h.mg.set_coarse_solver(pm.AmgSolver(h.operators[0]))
    n0 = h.mg.graph_replays()
    cycles(1)
    assert h.mg.graph_replays() == n0
    h.mg.set_coarse_solver(None)
    h.mg.set_graph(False)


def test_graph_cache_is_dropped_when_parts_are_swapped(pm):
    """a captured cycle holds the device pointers of the operators / smoothers / transfers / coarse
    solver it was captured with.  Swapping any of them with graph replay LEFT ENABLED must not replay the old
    objects: every setter drops the cache (and the handles are part of the cache key)."""
    h = pm.PoissonHierarchy(6, (1, 2, 4), kappa=2.0, cheb_its=3, warp=warp)
    import torch

    # the same levels with kappa = 3, on h's own layouts (set_operators checks the layout handles)
    kappa3 = torch.full((h.part.ncells,), 3.0, dtype=torch.float64, device="cuda")
    ops3 = []
    for P, lv, lay in zip(h.orders, h.levels, h.layouts):
        o = pm.MatFreeLaplacian(P, kappa3, lv.dofmap, h.xgeom, h.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, lay)
        o.compute_diag_inverse()
        ops3.append(o)
    b = h.rhs[-1]

    def cycles(mg, k=2):
        x = h.new_vector()
        x.set(0.0)
        for _ in range(k):
            mg.apply(b, x)
        return x.data_copy()

    # references, all eager
    e2 = cycles(h.mg)
    amg1, amg2 = pm.AmgSolver(h.operators[0], cycles=1), pm.AmgSolver(ops3[0], cycles=1)
    h.mg.set_coarse_solver(amg1)
    e_amg1 = cycles(h.mg)
    h.mg.set_coarse_solver(amg2)
    e_amg2 = cycles(h.mg)
    h.mg.set_coarse_solver(None)
    assert _relerr(e_amg2, e_amg1) > 1e-6

    h.mg.set_graph(True)
    y = h.new_vector()

    def replayed(k=2):  # same (rhs, y) pair every time: the cache key's vectors do not change
        y.set(0.0)
        for _ in range(k):
            h.mg.apply(b, y)
        return y.data_copy()

    assert _relerr(replayed(), e2) < 1e-13
    # swap the coarse AMG for one with the SAME parameters (so the same configuration hash but other matrices)
    h.mg.set_coarse_solver(amg1)
    assert _relerr(replayed(), e_amg1) < 1e-13
    h.mg.set_coarse_solver(amg2)
    assert _relerr(replayed(), e_amg2) < 1e-13
    h.mg.set_coarse_solver(None)
    # swap the operators (and with them smoothers' diagonals): kappa = 3 operators under the kappa = 2 multigrid
    h.mg.set_graph(False)
    h.mg.set_operators(ops3)
    e3 = cycles(h.mg)
    h.mg.set_operators(h.operators)
    h.mg.set_graph(True)
    assert _relerr(replayed(), e2) < 1e-13
    h.mg.set_operators(ops3)
    got = replayed()
    assert _relerr(got, e3) < 1e-13 and _relerr(got, e2) > 1e-6
    h.mg.set_operators(h.operators)
    h.mg.set_graph(False)


def test_cg_reports_a_poisoned_right_hand_side(pm):
    """a NaN right-hand side is an error, not "already converged"."""
    h = pm.PoissonHierarchy(4, (2,), kappa=2.0)
    cg = pm.CGSolver(h.layouts[-1])
    cg.set_max_iterations(5)
    cg.set_tolerance(1e-8)
    bad = h.new_vector()
    bad.set(float("nan"))
    x = h.new_vector()
    x.set(0.0)
    with pytest.raises(RuntimeError, match="not a non-negative finite"):
        cg.solve(h.operators[-1], x, bad)
    # zero right-hand side: converged at once, no error
    bad.set(0.0)
    assert cg.solve(h.operators[-1], x, bad) == 0


def test_errors(pm):
    part = pm.BoxPartition(2)
    lv = part.level(1)
    layout = pm.make_layout(lv)
    with pytest.raises(RuntimeError, match="Unsupported degree"):  # src/laplacian.hpp:346
        pm.MatFreeLaplacian(9, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
    with pytest.raises(ValueError):
        pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                            lv.bc_marker[:-1], layout)
    lv2 = part.level(2)
    with pytest.raises(RuntimeError, match="Incompatible vector sizes"):  # src/vector.hpp:343
        pm.inner_product(pm.Vector(layout), pm.Vector(pm.make_layout(lv2)))
    # empty cell list: nothing to do, output is zeroed
    op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap[:0], part.xgeom, part.geom_dofmap[:0], [], [], lv.bc_marker, layout)
    x, y = pm.Vector(layout), pm.Vector(layout)
    x.set(1.0)
    y.set(5.0)
    op(x, y)
    assert np.array_equal(y.data_copy(), np.zeros(lv.ndofs))


def test_full_size_properties(pm):
    """BASELINE config 2 size (64^3 hexes, p=4, 17M dofs): size-independent
    properties -- null space, symmetry, linearity, exact energy of a linear field."""
    part = pm.BoxPartition(64)
    lv = part.level(4)
    layout = pm.make_layout(lv)
    nobc = np.zeros_like(lv.bc_marker)
    op = pm.MatFreeLaplacian(4, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, nobc, layout)
    one, y = pm.Vector(layout), pm.Vector(layout)
    one.set(1.0)
    op(one, y)
    assert pm.norm(y, "linf") < 1e-11  # A 1 = 0
    g = torch.Generator(device="cuda").manual_seed(0)
    u, v, au, av = (pm.Vector(layout) for _ in range(4))
    u.data.copy_(torch.randn(lv.ndofs, generator=g, device="cuda", dtype=torch.float64))
    v.data.copy_(torch.randn(lv.ndofs, generator=g, device="cuda", dtype=torch.float64))
    op(u, au)
    op(v, av)
    vau, uav = pm.inner_product(v, au), pm.inner_product(u, av)
    assert abs(vau - uav) < 1e-11 * pm.norm(v) * pm.norm(au)  # symmetry
    w, aw = pm.Vector(layout), pm.Vector(layout)
    pm.axpy(w, 0.37, u, v)
    op(w, aw)
    pm.axpy(au, 0.37, au, av)  # 0.37 A u + A v
    pm.axpy(aw, -1.0, au, aw)
    assert pm.norm(aw) < 1e-12 * pm.norm(au)  # linearity
    xc = pm.Vector(layout)
    xc.data.copy_(torch.from_numpy(part.dof_coordinates(4)[:, 0].copy()))
    op(xc, y)
    assert abs(pm.inner_product(xc, y) - 2.0) < 1e-10  # u = x: u^T A u = kappa


def test_pcg_with_vcycle_preconditioner(pm):
    """BASELINE config 2 wording: CG preconditioned by the p-MG V-cycle (zero initial
    guess inside the preconditioner).  Not in the reference (its CGSolver hard-wires
    Jacobi, src/cg.hpp:154-161,192); checked against the oracle's PCG with the
    oracle's V-cycle."""
    from oracle import pmg_oracle as po

    n, orders, k = 6, (1, 2, 4), 3
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, warp=warp)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    for s_, e in zip(sm, h.eig_ranges):
        s_.eig_range = e
    cg = pm.CGSolver(h.layouts[-1])
    cg.set_max_iterations(50)
    cg.set_tolerance(1e-8)
    x = h.new_vector()
    x.set(0.0)
    its = cg.solve(h.operators[-1], x, h.rhs[-1], preconditioner=h.mg)
    ocg = po.CGSolver()
    ocg.set_max_iterations(50)
    ocg.set_tolerance(1e-8)
    xo = np.zeros_like(b)
    oits = ocg.solve(ops[-1], xo, b, precond=lambda r: mg.apply(r, np.zeros_like(r)))
    assert its == oits and its < 15
    assert _relerr(x.data_copy(), xo) < 1e-8
    # it really solves the system
    r = pm.Vector(h.layouts[-1])
    h.operators[-1](x, r)
    pm.axpy(r, -1.0, r, h.rhs[-1])
    assert pm.norm(r) < 1e-6 * pm.norm(h.rhs[-1])


@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 7, 8])
def test_affine_geometry_mode(pm, P):
    """Affine cells (here a sheared, stretched box: J constant, not diagonal): the
    constant-tensor apply equals the stored-G apply and the oracle; a mesh with a
    non-affine cell refuses the mode."""
    from oracle import pmg_oracle as po

    shear = np.array([[1.0, 0.2, 0.1], [0.0, 0.8, 0.3], [0.1, 0.0, 1.3]])
    lin = lambda x: x @ shear.T  # noqa: E731
    n = (4, 4, 8) if P <= 4 else (2, 2, 4)
    part = pm.BoxPartition(n, warp=lin)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
    assert op.is_affine()
    A = po.Laplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
    u = np.random.default_rng(P).standard_normal(lv.ndofs)
    x, y0, y1 = _vec(pm, layout, u), pm.Vector(layout), pm.Vector(layout)
    op(x, y0)
    op.set_geometry_mode("affine")
    op(x, y1)
    ref = A.apply(u)
    assert _relerr(y0.data_copy(), ref) < 1e-12
    assert _relerr(y1.data_copy(), ref) < 1e-12
    op.set_geometry_mode("stored")
    # trilinear (twisted) cells are not affine
    part2 = pm.BoxPartition((2, 2, 2), warp=twist)
    lv2 = part2.level(P)
    op2 = pm.MatFreeLaplacian(P, 2.0, lv2.dofmap, part2.xgeom, part2.geom_dofmap, lv2.lcells, lv2.bcells,
                              lv2.bc_marker, pm.make_layout(lv2))
    assert not op2.is_affine()
    with pytest.raises(RuntimeError, match="non-affine"):
        op2.set_geometry_mode("affine")


@pytest.mark.parametrize("P,pc", [(1, None), (2, 1), (3, None), (4, 2), (6, 3), (8, 4)])
def test_irregular_numbering(pm, P, pc):
    """What a dolfinx mesh looks like to the library: cells in arbitrary order, dofs
    renumbered arbitrarily, vertices off any tensor grid (so the patch builder takes its
    Morton-chunk path and has to close patches early).  Apply, inverse diagonal and the
    patched transfers against the oracle on the very same arrays."""
    from oracle import pmg_oracle as po

    rng = np.random.default_rng(1000 + P)
    n = (5, 4, 6) if P <= 4 else (3, 2, 3)
    part = pm.BoxPartition(n, warp=twist)
    lv = part.level(P)
    ncells, ndofs = part.ncells, lv.ndofs
    # jitter the interior vertices (cells stay valid hexes: the jitter is small against h)
    x = part.xgeom.copy()
    h = 1.0 / max(n)
    inner = np.all((x > 1e-9) & (x < 1 + 0.3), axis=1)
    x[inner] += 0.08 * h * rng.uniform(-1, 1, (int(inner.sum()), 3))
    cperm = rng.permutation(ncells)
    dperm = rng.permutation(ndofs)  # old dof -> new dof
    dofmap = dperm[lv.dofmap[cperm]].astype(np.int32)
    gdm = part.geom_dofmap[cperm]
    bc = np.zeros(ndofs, dtype=np.int8)
    bc[dperm] = lv.bc_marker
    kappa = rng.uniform(0.5, 3.0, ncells)
    # an arbitrary split into the two cell lists (the library must not care which is which)
    mask = rng.uniform(size=ncells) < 0.7
    lcells, bcells = np.nonzero(mask)[0].astype(np.int32), np.nonzero(~mask)[0].astype(np.int32)
    layout = pm.Layout(ndofs)
    op = pm.MatFreeLaplacian(P, kappa, dofmap, x, gdm, lcells, bcells, bc, layout)
    A = po.Laplacian(P, kappa, dofmap, x, gdm, bc)
    assert not op.is_affine()
    u = rng.standard_normal(ndofs)
    vx, vy = _vec(pm, layout, u), pm.Vector(layout)
    vy.set(9.0)
    op(vx, vy)
    assert _relerr(vy.data_copy(), A.apply(u)) < 1e-12
    op.compute_diag_inverse()
    d = pm.Vector(layout)
    op.get_diag_inverse(d)
    assert _relerr(d.data_copy(), A.diag_inverse()) < 1e-12
    if pc is None:
        return
    lvc = part.level(pc)
    cdperm = rng.permutation(lvc.ndofs)
    dmc = cdperm[lvc.dofmap[cperm]].astype(np.int32)
    Lc = pm.Layout(lvc.ndofs)
    ip = pm.Interpolator(pc, P, dmc, dofmap, lcells, bcells, Lc, layout, fine_operator=op)
    oi = po.Interpolator(pc, P, dmc, dofmap, lvc.ndofs, ndofs)
    uc = rng.standard_normal(lvc.ndofs)
    vc, vf = _vec(pm, Lc, uc), _vec(pm, layout, u)
    ip.interpolate_add(vc, vf)
    assert _relerr(vf.data_copy(), u + oi.interpolate(uc)) < 1e-13
    vc2 = pm.Vector(Lc)
    ip.reverse_interpolate(vx, vc2)
    assert _relerr(vc2.data_copy(), oi.reverse_interpolate(u)) < 1e-12


def test_vcycle_with_krylov_coarse_solver(pm):
    """set_coarse_solver (src/pmg.hpp:46,106-107): the coarsest level solved by CG from a zero
    initial guess (the reference: KSPCG, at most 60 iterations, src/amg.hpp:36-44 -- its hypre
    preconditioner is out of scope, here and in the oracle the CG is Jacobi-preconditioned)."""
    from oracle import pmg_oracle as po

    n, orders, k = 6, (1, 2, 4), 3
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, warp=warp)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    for s, e in zip(sm, h.eig_ranges):
        s.eig_range = e
    ccg = pm.CGSolver(h.layouts[0])
    ccg.set_max_iterations(12)  # a fixed count (rtol 0): identical arithmetic on both sides
    ccg.set_tolerance(0.0)
    h.mg.set_coarse_solver(ccg)

    def coarse(u0, b0):
        cg = po.CGSolver()
        cg.set_max_iterations(12)
        cg.set_tolerance(0.0)
        u0[:] = 0.0
        cg.solve(ops[0], u0, b0)

    mgo = po.MultigridPreconditioner(ops, sm, it, mesh.boundary_marker(orders[0]), coarse_solver=coarse)
    x = h.new_vector()
    x.set(0.0)
    xo = np.zeros_like(b)
    rn_cg = []
    for _ in range(3):
        rn = h.mg.apply(h.rhs[-1], x, verbose=True)
        xo = mgo.apply(b, xo, compute_rnorm=True)
        assert _relerr(x.data_copy(), xo) < 1e-9
        assert abs(rn - mgo.rnorm) < 1e-8 * mgo.rnorm
        rn_cg.append(rn)
    with pytest.raises(TypeError):
        h.mg.set_coarse_solver(object())  # no solve(x, b)
    wrong = pm.CGSolver(h.layouts[1])
    with pytest.raises(RuntimeError, match="coarsest layout"):
        h.mg.set_coarse_solver(wrong)
    # the reference's settings (60 iterations, rtol 1e-5) on a mesh whose coarse level is too big for
    # three smoother steps: the stationary iteration converges much faster with the Krylov coarse solve
    h2 = pm.PoissonHierarchy(24, (1, 2), kappa=2.0, cheb_its=2)
    x2 = h2.new_vector()
    x2.set(0.0)
    rs = [h2.mg.apply(h2.rhs[-1], x2, verbose=True) for _ in range(4)]
    c2 = pm.CGSolver(h2.layouts[0])
    c2.set_max_iterations(60)
    c2.set_tolerance(1e-5)
    h2.mg.set_coarse_solver(c2)
    x2.set(0.0)
    r60 = [h2.mg.apply(h2.rhs[-1], x2, verbose=True) for _ in range(4)]
    assert r60[-1] < 0.1 * rs[-1], (r60, rs)
    h2.mg.set_coarse_solver(None)
    x2.set(0.0)
    assert abs(h2.mg.apply(h2.rhs[-1], x2, verbose=True) - rs[0]) < 1e-12 * rs[0]  # NULL restores the smoother


def test_flexible_pcg_with_krylov_coarse_solver(pm):
    """Outer CG with the Polak-Ribiere beta (pmg_cg_set_flexible) and a V-cycle whose coarsest
    level is solved by an inner CG -- the preconditioner is then not a fixed linear operator.
    Against the oracle's flexible PCG with the same inner CG (fixed iteration count)."""
    from oracle import pmg_oracle as po

    n, orders, k = 8, (1, 2, 4), 2
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, warp=warp)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    for s_, e in zip(sm, h.eig_ranges):
        s_.eig_range = e
    ccg = pm.CGSolver(h.layouts[0])
    ccg.set_max_iterations(8)
    ccg.set_tolerance(0.0)
    h.mg.set_coarse_solver(ccg)

    def coarse(u0, b0):
        c = po.CGSolver()
        c.set_max_iterations(8)
        c.set_tolerance(0.0)
        u0[:] = 0.0
        c.solve(ops[0], u0, b0)

    mgo = po.MultigridPreconditioner(ops, sm, it, mesh.boundary_marker(orders[0]), coarse_solver=coarse)
    cg = pm.CGSolver(h.layouts[-1])
    cg.set_max_iterations(40)
    cg.set_tolerance(1e-9)
    cg.set_flexible(True)
    x = h.new_vector()
    x.set(0.0)
    its = cg.solve(h.operators[-1], x, h.rhs[-1], preconditioner=h.mg)
    ocg = po.CGSolver()
    ocg.set_max_iterations(40)
    ocg.set_tolerance(1e-9)
    xo = np.zeros_like(b)
    oits = ocg.solve(ops[-1], xo, b, precond=lambda r: mgo.apply(r, np.zeros_like(r)), flexible=True)
    assert its == oits and its < 15
    assert _relerr(x.data_copy(), xo) < 1e-8
    r = pm.Vector(h.layouts[-1])
    y = pm.Vector(h.layouts[-1])
    h.operators[-1](x, y)
    pm.axpy(r, -1.0, y, h.rhs[-1])
    assert pm.norm(r) < 1e-7 * pm.norm(h.rhs[-1])  # the recurrence residual is honest


def test_apply_parity_random_small_meshes(pm):
    """Seeded sweep over odd little meshes (single cells, one-cell-thick slabs, sizes that leave
    partial patches and partial wave items in every direction) and all degrees, with an arbitrary
    split into the two cell lists and a random Dirichlet marker."""
    from oracle import pmg_oracle as po

    rng = np.random.default_rng(2024)
    shapes = [(1, 1, 1), (1, 1, 9), (9, 1, 1), (2, 3, 1), (1, 5, 2), (3, 3, 3), (5, 2, 7), (4, 4, 9)]
    for case in range(24):
        P = int(rng.integers(1, 9))
        n = shapes[case % len(shapes)] if P <= 4 else shapes[case % 6]
        part = pm.BoxPartition(n, warp=twist)
        lv = part.level(P)
        bc = (rng.uniform(size=lv.ndofs) < 0.15).astype(np.int8)
        kappa = rng.uniform(0.5, 2.0, part.ncells)
        mask = rng.uniform(size=part.ncells) < 0.6
        lcells = np.nonzero(mask)[0].astype(np.int32)
        bcells = np.nonzero(~mask)[0].astype(np.int32)
        layout = pm.Layout(lv.ndofs)
        op = pm.MatFreeLaplacian(P, kappa, lv.dofmap, part.xgeom, part.geom_dofmap, lcells, bcells, bc, layout)
        A = po.Laplacian(P, kappa, lv.dofmap, part.xgeom, part.geom_dofmap, bc)
        u = rng.standard_normal(lv.ndofs)
        x, y = _vec(pm, layout, u), pm.Vector(layout)
        y.set(3.0)
        op(x, y)
        assert _relerr(y.data_copy(), A.apply(u)) < 1e-12, (case, P, n)


# every degree, on meshes that give at least two full patches per colour (patch shapes:
# patches.hpp patch_shape) -- the coloured launches are the path the bench times
@pytest.mark.parametrize("P,n", [(1, (8, 8, 32)), (2, (8, 8, 32)), (3, (4, 4, 32)), (4, (4, 4, 32)), (5, (4, 4, 14)),
                                 (6, (4, 4, 8)), (7, (4, 4, 12)), (8, (2, 2, 12))])
def test_merged_and_coloured_launches_agree(pm, P, n):
    """The same operator built with the interior colours as separate launches (plain stores) and
    merged into one launch (atomics): same result, and both equal the oracle."""
    from oracle import c_oracle as co

    part = pm.BoxPartition(n, warp=twist)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    A = co.CLevel(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
    u = np.random.default_rng(7).standard_normal(lv.ndofs)
    x = _vec(pm, layout, u)
    got = {}
    try:
        for name, below in (("coloured", 0), ("merged", 10**12)):
            pm.set_merge_threshold(below)
            op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                                     lv.bc_marker, layout)
            y = pm.Vector(layout)
            y.set(1.0)
            op(x, y)
            first = y.data_copy()
            op(x, y)  # second application onto the first one's output: nothing may carry over
            got[name] = (first, op.launches_per_apply(), y.data_copy())
    finally:
        pm.set_merge_threshold(-1)
    assert got["coloured"][1] >= 8 and got["merged"][1] == 1
    ref = A.apply(u)
    for name in ("coloured", "merged"):
        assert _relerr(got[name][0], ref) < 1e-12, name
        assert _relerr(got[name][2], ref) < 1e-12, name
    # no global atomics on the coloured path; inside a patch the cell sums meet in LDS in whatever
    # order the wavefronts arrive, so two runs agree to rounding, not bit for bit
    assert _relerr(got["coloured"][0], got["coloured"][2]) < 1e-14


# the interior of a large level runs as two halves on two streams (include/pmg_amd.h, pmg_laplacian_apply_streams);
# PMG_APPLY_STREAMS=2 asks for that on the small meshes a test can check against the oracle
@pytest.mark.parametrize("P,n,shuffle", [(1, (16, 8, 32), False), (2, (16, 8, 32), False), (3, (8, 8, 32), False),
                                         (4, (8, 6, 32), False), (4, (6, 12, 16), False), (6, (4, 4, 32), False),
                                         (8, (8, 2, 12), False), (4, (8, 4, 16), True), (2, (8, 8, 32), True)])
def test_two_stream_halves_agree_with_oracle(pm, P, n, shuffle, monkeypatch):
    """Coloured launches split into two halves on two streams: same vector as the oracle, twice in a row (the event
    that orders the dofs along the cut), under a stream capture as well; a mesh whose cells arrive in random order
    (patches from Morton chunks, colours that do not separate the two sides of the cut) falls back to one sequence or
    splits -- either way the result is the oracle's."""
    from oracle import c_oracle as co

    # (a separable stretch keeps the centroids on a tensor grid, which is what makes tensor-block patches; the
    # shuffled cases use the twisted mesh, whose patches are Morton chunks)
    part = pm.BoxPartition(n, warp=twist if shuffle else (lambda x: x + 0.05 * np.sin(2.0 * np.pi * x)))
    lv = part.level(P)
    layout = pm.make_layout(lv)
    lcells = lv.lcells
    if shuffle:
        lcells = np.ascontiguousarray(np.random.default_rng(5).permutation(lcells))
    A = co.CLevel(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
    u = np.random.default_rng(11).standard_normal(lv.ndofs)
    x = _vec(pm, layout, u)
    ref = A.apply(u)
    monkeypatch.setenv("PMG_APPLY_STREAMS", "2")
    try:
        pm.set_merge_threshold(0)
        op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lcells, lv.bcells, lv.bc_marker,
                                 layout)
        monkeypatch.setenv("PMG_APPLY_STREAMS", "0")
        one = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lcells, lv.bcells, lv.bc_marker,
                                  layout)
    finally:
        pm.set_merge_threshold(-1)
    assert one.apply_streams() == 1
    if not shuffle:
        assert op.apply_streams() == 2
    y = pm.Vector(layout)
    for rep in range(3):
        y.set(float(rep))
        op(x, y)
        assert _relerr(y.data_copy(), ref) < 1e-12, rep
    # captured: the fork / join become two branches of the graph
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        op(x, y)
        g = torch.cuda.CUDAGraph()
        y.set(-1.0)
        with torch.cuda.graph(g, stream=s):
            op(x, y)
        y.set(5.0)
        g.replay()
        g.replay()
    torch.cuda.synchronize()
    assert _relerr(y.data_copy(), ref) < 1e-12
    # diagonal and transfers built on the same patches
    op.compute_diag_inverse()
    one.compute_diag_inverse()
    d2, d1 = pm.Vector(layout), pm.Vector(layout)
    op.get_diag_inverse(d2)
    one.get_diag_inverse(d1)
    assert _relerr(d2.data_copy(), d1.data_copy()) < 1e-13


# the interior of a large degree-4 level as chains of patches, one persistent workgroup per chain (include/pmg_amd.h,
# "Chain form"); PMG_CHAIN=2 builds the chains on the small meshes a test can check against the oracle
@pytest.mark.parametrize("n,wf,bc", [((4, 4, 64), None, True),      # z-chains of eight full patches
                                    ((8, 6, 32), "stretch", True),  # the y axis has the fewest patches: chains along y
                                    ((6, 12, 16), "stretch", False),
                                    ((4, 4, 19), None, True),       # patches cut short (28 and 24 cells), chains of three
                                    ((5, 3, 19), None, True),       # thin blocks folded together: no grid of patches
                                    ((2, 2, 40), "twist", True),    # one column, trilinear cells
                                    ((8, 8, 8), None, True)])       # chains of ONE patch (eight colours: not taken)
def test_chain_form_agrees_with_oracle(pm, n, wf, bc, monkeypatch):
    """Degree 4: the chain form of the interior launches gives the oracle's vector -- twice in a row (nothing carries
    over in y or in the workgroup's buffers), equal to the patch launches to rounding, under a stream capture as well,
    with Dirichlet rows (y = x) written once."""
    from oracle import c_oracle as co

    P = 4
    warpf = {None: None, "stretch": (lambda x: x + 0.05 * np.sin(2.0 * np.pi * x)), "twist": twist}[wf]
    part = pm.BoxPartition(n, warp=warpf)
    lv = part.level(P)
    bcm = lv.bc_marker if bc else np.zeros_like(lv.bc_marker)
    layout = pm.make_layout(lv)
    A = co.CLevel(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, bcm)
    u = np.random.default_rng(23).standard_normal(lv.ndofs)
    x = _vec(pm, layout, u)
    ref = A.apply(u)
    monkeypatch.setenv("PMG_CHAIN", "2")
    try:
        pm.set_merge_threshold(0)
        op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, bcm, layout)
    finally:
        pm.set_merge_threshold(-1)
    if n in ((8, 8, 8), (5, 3, 19)) or wf == "twist":
        # a 4 x 4 x 1 grid of patches strung along z gives chains of one patch in eight colours -- no fewer launches
        # than the patch colours, so the level keeps those; the twisted mesh's patches are Morton chunks
        if not op.chain_available():
            with pytest.raises(pm._lib.PmgError):
                op.set_chain_form(True)
            return
    assert op.chain_available() and op.chain_form()
    colours = op.launches_per_apply()
    assert colours <= 4
    y = pm.Vector(layout)
    for rep in range(3):
        y.set(float(rep) - 1.0)
        op(x, y)
        assert _relerr(y.data_copy(), ref) < 1e-12, rep
    chained = y.data_copy()
    op.set_chain_form(False)
    assert op.launches_per_apply() >= 8
    y.set(3.0)
    op(x, y)
    assert _relerr(y.data_copy(), ref) < 1e-12
    assert _relerr(chained, y.data_copy()) < 1e-13
    op.set_chain_form(True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        op(x, y)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            op(x, y)
        y.set(5.0)
        g.replay()
        g.replay()
    torch.cuda.synchronize()
    assert _relerr(y.data_copy(), ref) < 1e-12
    # kappa is read in every application (the caller's array may change between two)
    op2 = pm.MatFreeLaplacian(P, 0.5, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, bcm, layout)
    y2 = pm.Vector(layout)
    op2(x, y2)
    free = ~bcm.astype(bool)
    assert _relerr(4.0 * y2.data_copy()[free], chained[free]) < 1e-12
