"""Cell-local node order at the boundary (include/pmg_amd.h, "cell-local node order").

The reference's dofmaps come from a basix tensor-product element (examples/pmg/main.cpp:83-87)
and its tables from basix (src/laplacian.hpp:302-317, src/interpolate.hpp:118): along every
direction the nodes are numbered vertex 0, vertex 1, interior left to right.  The library's
kernels number them by ascending coordinate.  These tests hand the library dofmaps and tables
exactly as a dolfinx caller holds them (``*_ordered`` entry points, PMG_NODES_ENDPOINTS_FIRST)
and require the GLOBAL vectors of the oracle -- which never sees that order -- to 1e-12 (apply,
transfers) and 1e-10 (V-cycle), P = 1 ... 8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def twist(x):
    y = x.copy()
    y[:, 0] += 0.12 * x[:, 1] * x[:, 2]
    y[:, 1] += 0.10 * x[:, 0] * x[:, 2] + 0.05 * x[:, 0] * x[:, 1] * x[:, 2]
    y[:, 2] += 0.08 * x[:, 0] * x[:, 1]
    return y


def warp(x):
    return x + 0.03 * np.sin(3.0 * x[:, [1, 2, 0]])


@pytest.fixture(scope="module")
def pm(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pmg_dolfinx_amd as pm

    torch.cuda.set_device(0)
    return pm


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _vec(pm, layout, a):
    v = pm.Vector(layout)
    v.data.copy_(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)))
    return v


def _caller_tables(po, pm, P, perm1d):
    """dphi_geometry [3][nq][8] and G_weights [nq] with the quadrature points in the caller's order
    (what basix's tabulate / make_quadrature return for perm1d = basix)."""
    dphi, w3 = po.geometry_tables(P)
    p3 = pm.cell_permutation(perm1d)  # caller q -> ascending q
    return np.ascontiguousarray(dphi[:, p3, :]), np.ascontiguousarray(w3[p3])


@pytest.mark.parametrize("tables", [False, True])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 7, 8])
def test_operator_from_basix_ordered_arrays(pm, P, tables):
    """Apply, geometry tensor, inverse diagonal and load vector of an operator built from an endpoints-first dofmap
    (and, with ``tables``, from the caller's own coordinate-element tabulation and weights in that order)."""
    from oracle import pmg_oracle as po

    n = (3, 2, 4) if P > 4 else (5, 4, 3)
    part = pm.BoxPartition(n, warp=twist)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    perm = pm.basix_node_permutation(P)
    assert list(perm) == list(pm.node_permutation("basix", P))
    dm_basix = pm.dofmap_in_node_order(lv.dofmap, perm)
    kw = {}
    if tables:
        kw["dphi_geometry"], kw["G_weights"] = _caller_tables(po, pm, P, perm)
    op = pm.MatFreeLaplacian(P, 2.0, dm_basix, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker,
                             layout, node_order="basix", **kw)
    A = po.Laplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
    u = np.random.default_rng(40 + P).standard_normal(lv.ndofs)
    x, y = _vec(pm, layout, u), pm.Vector(layout)
    y.set(-2.0)
    op(x, y)
    assert _relerr(y.data_copy(), A.apply(u)) < 1e-12
    # the tensor comes back indexed by the CALLER's quadrature-point numbers
    p3 = pm.cell_permutation(perm)
    assert _relerr(op.geometry().cpu().numpy(), A.G[:, p3, :]) < 1e-13
    op.compute_diag_inverse()
    d = pm.Vector(layout)
    op.get_diag_inverse(d)
    assert _relerr(d.data_copy(), A.diag_inverse()) < 1e-12
    # load vector: nodal values by dof number, no cell-local order involved
    f = np.random.default_rng(7).standard_normal(lv.ndofs)
    b = pm.Vector(layout)
    op.assemble_rhs(_vec(pm, layout, f), b)
    ref = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker,
                              layout)
    b2 = pm.Vector(layout)
    ref.assemble_rhs(_vec(pm, layout, f), b2)
    assert _relerr(b.data_copy(), b2.data_copy()) < 1e-13
    if P >= 2:
        # the order matters: the same array declared ascending gives another operator
        wrong = pm.MatFreeLaplacian(P, 2.0, dm_basix, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                                    lv.bc_marker, layout)
        wrong(x, y)
        assert _relerr(y.data_copy(), A.apply(u)) > 1e-3


@pytest.mark.parametrize("patched", [False, True])
@pytest.mark.parametrize("pc,pf", [(1, 2), (2, 4), (1, 3), (3, 6), (4, 8), (2, 3), (1, 5), (3, 7)])
def test_transfers_from_basix_ordered_dofmaps(pm, pc, pf, patched):
    from oracle import pmg_oracle as po

    n = (3, 2, 2)
    part = pm.BoxPartition(n, warp=warp)
    lc, lf = part.level(pc), part.level(pf)
    Lc, Lf = pm.make_layout(lc), pm.make_layout(lf)
    dmc = pm.dofmap_in_node_order(lc.dofmap, pm.basix_node_permutation(pc))
    dmf = pm.dofmap_in_node_order(lf.dofmap, pm.basix_node_permutation(pf))
    fop = None
    if patched:
        fop = pm.MatFreeLaplacian(pf, 2.0, dmf, part.xgeom, part.geom_dofmap, lf.lcells, lf.bcells, lf.bc_marker, Lf,
                                  node_order="basix")
    ip = pm.Interpolator(pc, pf, dmc, dmf, lf.lcells, lf.bcells, Lc, Lf, fine_operator=fop, node_order="basix")
    oi = po.Interpolator(pc, pf, lc.dofmap, lf.dofmap, lc.ndofs, lf.ndofs)
    rng = np.random.default_rng(pc * 10 + pf)
    uc, uf = rng.standard_normal(lc.ndofs), rng.standard_normal(lf.ndofs)
    vc, vf = _vec(pm, Lc, uc), pm.Vector(Lf)
    ip.interpolate(vc, vf)
    assert _relerr(vf.data_copy(), oi.interpolate(uc)) < 1e-13
    vf2, vc2 = _vec(pm, Lf, uf), pm.Vector(Lc)
    vc2.set(3.0)
    ip.reverse_interpolate(vf2, vc2)
    assert _relerr(vc2.data_copy(), oi.reverse_interpolate(uf)) < 1e-12
    if patched:
        vf3 = _vec(pm, Lf, uf)
        ip.interpolate_add(vc, vf3)
        assert _relerr(vf3.data_copy(), uf + oi.interpolate(uc)) < 1e-13


@pytest.mark.parametrize("orders,n", [((1, 2, 4), 4), ((1, 3, 6), 3), ((1, 2, 4, 8), 3), ((1, 5), (2, 2, 7)),
                                      ((3, 7), (2, 2, 3))])
def test_vcycle_from_basix_ordered_dofmaps(pm, orders, n):
    """The whole hierarchy wired from endpoints-first dofmaps: eigenvalue estimates, load vector and three
    V-cycles against the oracle (which works on ascending dofmaps throughout)."""
    from oracle import pmg_oracle as po

    k = 3
    h = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, warp=warp, node_order="basix")
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
    for got, ref in zip(h.eig_ranges, eigs):
        assert abs(got[1] - ref[1]) < 1e-8 * ref[1]
    for s, e in zip(sm, h.eig_ranges):
        s.eig_range = e
    assert _relerr(h.rhs[-1].data_copy(), b) < 1e-12
    x = h.new_vector()
    x.set(0.0)
    xo = np.zeros_like(b)
    for cyc in range(3):
        rn = h.mg.apply(h.rhs[-1], x, verbose=True)
        xo = mg.apply(b, xo, compute_rnorm=True)
        assert _relerr(x.data_copy(), xo) < 1e-10
        assert abs(rn - mg.rnorm) < 1e-9 * max(mg.rnorm, 1e-30) + 1e-12


def test_custom_permutation_and_errors(pm):
    """PMG_NODES_CUSTOM with an arbitrary 1-D permutation; a non-permutation is refused."""
    from oracle import pmg_oracle as po

    P, n = 4, (3, 3, 2)
    part = pm.BoxPartition(n, warp=twist)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    perm = np.array([3, 0, 4, 1, 2], dtype=np.int32)
    dm = pm.dofmap_in_node_order(lv.dofmap, perm)
    op = pm.MatFreeLaplacian(P, 2.0, dm, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout,
                             node_order="custom", perm1d=perm)
    A = po.Laplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
    u = np.random.default_rng(3).standard_normal(lv.ndofs)
    x, y = _vec(pm, layout, u), pm.Vector(layout)
    op(x, y)
    assert _relerr(y.data_copy(), A.apply(u)) < 1e-12
    with pytest.raises(RuntimeError, match="not a permutation"):
        pm.MatFreeLaplacian(P, 2.0, dm, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout,
                            node_order="custom", perm1d=np.array([0, 1, 2, 3, 3], dtype=np.int32))
    with pytest.raises(RuntimeError, match="unknown order"):
        pm.MatFreeLaplacian(P, 2.0, dm, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout,
                            node_order=7)
