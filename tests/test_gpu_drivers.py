"""The C++ drivers over the adapter header (examples/, include/pmg_amd.hpp) against
the oracle: the same runs as the reference's examples/mat_free/main.cpp and
examples/pmg/main.cpp, executed as separate processes that link libpmg_amd.so."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pmg-dolfinx_amd", "bin")


def run(exe, *args):
    path = os.path.join(BIN, exe)
    assert os.path.exists(path), f"{path} missing: run __graft_entry__.build()"
    r = subprocess.run([path, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def grab(pattern, text):
    return [float(v) for v in re.findall(pattern, text)]


@pytest.mark.parametrize("n,P", [(16, 1), (5, 4), (3, 7)])
def test_mat_free_driver(built, n, P):
    from oracle import pmg_oracle as po

    out = run("mat_free_main", "--n", n, "--degree", P, "--nreps", 5, *(["--mat_comp"] if P == 1 else []))
    mesh = po.BoxMesh(n)
    A = po.Laplacian(P, 2.0, mesh.dofmap(P), mesh.xgeom, mesh.geom_dofmap, mesh.boundary_marker(P))
    u = np.ones(A.ndofs)
    y = A.apply(u)
    (nu,) = grab(r"Norm of u = (\S+)", out)
    (ny,) = grab(r"Norm of y = (\S+)", out)
    assert abs(nu - np.linalg.norm(u)) < 1e-12 * nu
    assert abs(ny - np.linalg.norm(y)) < 1e-12 * ny  # tolerance: 1e-12 relative per apply
    if P == 1:
        (err,) = grab(r"Norm of error = (\S+)", out)
        assert err < 1e-12  # the P=1 operator is the 7-point stencil (KAT 1)


def test_pmg_driver(built):
    from oracle import pmg_oracle as po

    n, orders, k, cycles = 6, (1, 2, 4), 3, 4
    out = run("pmg_main", "--n", n, "--orders", ",".join(map(str, orders)), "--smoother-its", k, "--cycles", cycles,
              "--pcg")
    lam = grab(r"Eigenvalues level \d+: \S+ - (\S+)", out)
    rn = grab(r"Cycle \d+: residual norm = (\S+)", out)
    (nb,) = grab(r"Norm of b = (\S+)", out)
    (nx,) = grab(r"Norm of x = (\S+)\n", out)[:1]
    assert len(lam) == len(orders) and len(rn) == cycles

    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k)
    for got, ref in zip(lam, eigs):
        assert abs(1.1 * got - ref[1]) < 1e-8 * ref[1]
    for s, l in zip(sm, lam):  # same smoother bounds, so only the cycle arithmetic is compared
        s.eig_range = (0.1 * l, 1.1 * l)
    assert abs(nb - np.linalg.norm(b)) < 1e-12 * nb
    x = np.zeros_like(b)
    for c in range(cycles):
        x = mg.apply(b, x, compute_rnorm=True)
        assert abs(rn[c] - mg.rnorm) < 1e-8 * mg.rnorm + 1e-13, (c, rn[c], mg.rnorm)
    assert abs(nx - np.linalg.norm(x)) < 1e-10 * nx  # tolerance: 1e-10 after full V-cycles
    assert rn[-1] < 1e-2 * rn[0]

    m = re.search(r"PCG with V-cycle preconditioner: (\d+) iterations, \|b - A x\| / \|b\| = (\S+), "
                  r"Norm of x = (\S+)", out)
    its = int(m.group(1))
    assert 1 <= its <= 15 and float(m.group(2)) < 1e-6
    # converged solution of A x = b, independent of the path taken
    cg = po.CGSolver()
    cg.set_max_iterations(2000)
    cg.set_tolerance(1e-13)
    xs = np.zeros_like(b)
    cg.solve(ops[-1], xs, b)
    assert abs(float(m.group(3)) - np.linalg.norm(xs)) < 1e-6 * np.linalg.norm(xs)


def test_pmg_driver_with_basix_ordered_dofmaps(built):
    """--node-order basix: the C++ adapter hands the library dofmaps in basix's cell-local node order (endpoints first
    per direction: what dolfinx gives the reference, examples/pmg/main.cpp:83-87) through
    pmg_laplacian_create_ordered / pmg_interpolator_create_ordered; the run reproduces the ascending run's numbers."""
    args = ("--n", 6, "--orders", "1,2,4", "--smoother-its", 3, "--cycles", 4, "--pcg")
    ref, out = run("pmg_main", *args), run("pmg_main", *args, "--node-order", "basix")
    for pat in (r"Eigenvalues level \d+: \S+ - (\S+)", r"Cycle \d+: residual norm = (\S+)", r"Norm of b = (\S+)"):
        a, b = grab(pat, ref), grab(pat, out)
        assert len(a) == len(b) > 0 and all(abs(x - y) < 1e-9 * abs(x) for x, y in zip(a, b)), (pat, a, b)


def test_driver_errors(built):
    path = os.path.join(BIN, "mat_free_main")
    r = subprocess.run([path, "--degree", "9"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Unsupported degree" in r.stderr  # src/laplacian.hpp:346


def test_cpp_adapter_selftest(built):
    """Vector algebra, halo scatter through a C++ exchange callback, compute_boundary_cells and
    the reference's error behaviour, checked inside the C++ program itself."""
    out = run("selftest_main")
    assert "adapter selftest passed" in out


def test_pmg_driver_coarse_cg(built):
    """--coarse-cg: the role of the reference driver's --amg (a Krylov coarse solver in the cycle)."""
    args = ("--n", 24, "--orders", "1,2", "--smoother-its", 2, "--cycles", 4)
    plain = grab(r"Cycle \d+: residual norm = (\S+)", run("pmg_main", *args))
    krylov = grab(r"Cycle \d+: residual norm = (\S+)", run("pmg_main", *args, "--coarse-cg"))
    assert len(plain) == len(krylov) == 4
    assert krylov[-1] < 0.1 * plain[-1]


def test_pmg_driver_amg_and_native_communicator(built, tmp_path):
    """--amg (the reference driver's flag, examples/pmg/main.cpp:331-335): CG + the library's AMG on the
    degree-1 level; --amg-cycles: the same hierarchy as stationary cycles; --native-comm: the same run
    with every scatter and reduction going through the RCCL communicator (one rank); --output."""
    args = ("--n", 24, "--orders", "1,2,4", "--smoother-its", 3, "--cycles", 5)
    plain = grab(r"Cycle \d+: residual norm = (\S+)", run("pmg_main", *args))
    out = run("pmg_main", *args, "--amg", "--pcg", "--random-rhs")
    amg = grab(r"Cycle \d+: residual norm = (\S+)", out)
    assert "AMG coarse solver:" in out
    assert all(amg[i + 1] < 0.2 * amg[i] for i in range(4)), amg  # contraction per cycle with a solved coarse level
    assert amg[-1] < 1e-2 * plain[-1]
    m = re.search(r"PCG with V-cycle preconditioner: (\d+) iterations, \|b - A x\| / \|b\| = ([0-9.e+-]+)", out)
    assert int(m.group(1)) <= 12 and float(m.group(2)) < 1e-6
    stat = grab(r"Cycle \d+: residual norm = (\S+)", run("pmg_main", *args, "--amg-cycles", 2))
    assert all(stat[i + 1] < 0.2 * stat[i] for i in range(4)), stat
    # the RCCL communicator path gives the same numbers as the plain single-rank path
    vtk = tmp_path / "u.vtk"
    nat = run("pmg_main", *args, "--native-comm", "--id-file", tmp_path / "id", "--output", vtk)
    got = grab(r"Cycle \d+: residual norm = (\S+)", nat)
    assert len(got) == len(plain) and all(abs(a - b) < 1e-9 * b for a, b in zip(got, plain))
    text = vtk.read_text().split("\n")
    assert text[0].startswith("# vtk") and any(line.startswith("POINT_DATA") for line in text)


def test_mat_free_driver_geometry_batching(built):
    """--batch_size (examples/mat_free/main.cpp:34-50, src/laplacian.hpp:383-396): G recomputed batch by
    batch in every apply -- same result, a fraction of the memory."""
    full = run("mat_free_main", "--n", 12, "--degree", 3, "--nreps", 3)
    bat = run("mat_free_main", "--n", 12, "--degree", 3, "--nreps", 3, "--batch_size", 200)
    (y0,) = grab(r"Norm of y = (\S+)", full)
    (y1,) = grab(r"Norm of y = (\S+)", bat)
    assert abs(y0 - y1) < 1e-13 * y0  # the same tensor values (sums meet in LDS in arrival order)
    (m0,) = grab(r"Geometry tensor held: (\S+) MB", full)
    (m1,) = grab(r"Geometry tensor held: (\S+) MB", bat)
    assert m1 < 0.2 * m0 and "recomputed" in bat


def test_cg_driver(built):
    """examples/cg: Jacobi-CG eigenvalue estimate, then Chebyshev from a non-zero guess on the load
    vector with a non-zero Dirichlet value (lifting), against the oracle."""
    from oracle import pmg_oracle as po

    n, P, g_bc = 5, 3, 1.3
    out = run("cg_main", "--n", n, "--degree", P)
    mesh = po.BoxMesh(n)
    marker = mesh.boundary_marker(P)
    A = po.Laplacian(P, 2.0, mesh.dofmap(P), mesh.xgeom, mesh.geom_dofmap, marker)
    A0 = po.Laplacian(P, 2.0, mesh.dofmap(P), mesh.xgeom, mesh.geom_dofmap, np.zeros_like(marker))
    bc = marker.astype(bool)

    cg = po.CGSolver()
    cg.set_max_iterations(20)
    cg.set_tolerance(1e-6)
    cg.store_coefficients(True)
    its = cg.solve(A, np.zeros(A.ndofs), np.ones(A.ndofs))
    eig = cg.compute_eigenvalues()
    (got_its,) = grab(r"Number of iterations (\d+)", out)
    lo, hi = re.findall(r"Computed eigs = \(([0-9.e+-]+), ([0-9.e+-]+)\)", out)[0]
    assert int(got_its) == its
    assert abs(float(lo) - eig[0]) < 1e-8 * eig[-1] and abs(float(hi) - eig[-1]) < 1e-10 * eig[-1]

    c = mesh.dof_coordinates(P)
    f = 1000 * np.exp(-((c[:, 0] - 0.5) ** 2 + (c[:, 1] - 0.5) ** 2) / 0.02)
    wdet = A0.w3[None, :] * A0.detJ  # L = inner(f, v) * dx under the GLL rule: lumped mass (no kappa)
    b = np.bincount(A0.dofmap.ravel(), weights=(wdet * f[A0.dofmap]).ravel(), minlength=A0.ndofs)
    g = np.where(bc, g_bc, 0.0)
    b = b - A0.apply(g)  # apply_lifting
    b[bc] = g_bc  # set_bc
    (nb,) = grab(r"Norm of b = (\S+)", out)
    assert abs(nb - np.linalg.norm(b)) < 1e-12 * nb

    x0 = np.where(bc, g_bc, 1.0)
    got = dict((int(k), float(v)) for k, v in re.findall(r"Chebyshev iteration (\d+): residual norm = (\S+)", out))
    assert set(got) == {0, 1, 2, 5, 10, 20, 30}
    for k, rn in sorted(got.items()):
        x = x0.copy()
        if k:
            po.Chebyshev((0.1 * eig[-1], 1.1 * eig[-1]), k).solve(A, x, b)
        ref = np.linalg.norm(b - A.apply(x))
        assert abs(rn - ref) < 1e-10 * got[0], (k, rn, ref)
    (nx,) = grab(r"Norm of x = (\S+)", out)
    assert abs(nx - np.linalg.norm(x)) < 1e-10 * nx


def test_vector_update_driver(built, tmp_path):
    """examples/vector-update: 100 x {scatter begin; norm; axpy; scatter end}; norms in closed form
    (x = rank + i on every owned dof); once without and once through the RCCL communicator."""
    for extra in ([], ["--native-comm", "--id-file", str(tmp_path / "id")]):
        out = run("vector_update_main", "--ndofs", 20000, *extra)
        (nd,) = grab(r"Number of dofs-global : (\d+)", out)
        vals = grab(r"Dot value: (\S+)", out)
        assert len(vals) == 100
        for i, v in enumerate(vals):
            assert abs(v - i * np.sqrt(nd)) < 1e-12 * max(v, 1.0)
        assert "Ghost check" in out
        (nx,) = grab(r"Norm of x = (\S+)", out)
        assert abs(nx - 100 * np.sqrt(nd)) < 1e-10 * nx


def test_pmg_driver_graph_replay(built, tmp_path):
    """--graph: the timed cycles run as hipGraph replays, also on a layout with the RCCL communicator (and with
    --halo windows: the adapter's window bootstrap over the communicator, here without neighbours)."""
    for extra in ([], ["--native-comm", "--id-file", str(tmp_path / "id")],
                  ["--native-comm", "--halo", "windows", "--id-file", str(tmp_path / "id")]):
        out = run("pmg_main", "--n", 6, "--orders", "1,2,4", "--cycles", 2, "--graph", *extra)
        assert "(hipGraph replays)" in out
        ref = run("pmg_main", "--n", 6, "--orders", "1,2,4", "--cycles", 2, *extra)
        (a,), (b,) = grab(r"Cycle 2: residual norm = (\S+)", out), grab(r"Cycle 2: residual norm = (\S+)", ref)
        assert abs(a - b) < 1e-10 * b  # atomic-order noise of the merged launches


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_the_box_has():
    """`bench.py --gpus N` on a box with fewer devices exits non-zero with a message and
    prints no line (never a 1-rank measurement labelled N GPUs)."""
    import subprocess
    import sys

    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert f"--gpus {n}" in r.stderr and "refusing" in r.stderr


@pytest.mark.parametrize("dims", ["1,1,2", "1,2,2"])
def test_pmg_driver_ranks_share_the_gpu_through_the_window_communicator(built, dims):
    """examples/pmg/run_ranks.sh with --comm windows: the C++ driver as 2 and 4 PROCESSES, one brick each, on the one
    GPU of the box (RCCL refuses two ranks on a device; the communicator made of windows does not care): the file
    bootstrap of the window handles, halo windows, the all-reduces of CG and of the replicated AMG.  Same global
    problem as the single-process run, so the same residuals cycle by cycle, and hipGraph replays on top."""
    script = os.path.join(ROOT, "examples", "pmg", "run_ranks.sh")
    args = ["--n", "8", "--orders", "1,2,4", "--smoother-its", "3", "--cycles", "4"]

    def ranks(*extra):
        r = subprocess.run(["bash", script, dims, *args, "--comm", "windows", *extra], capture_output=True, text=True,
                           timeout=300, env={**os.environ, "PMG_WINDOW_TIMEOUT_MS": "20000"})
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    for extra in ([], ["--amg-cycles", "2", "--amg-setup", "gathered"], ["--amg-cycles", "2"]):
        one = grab(r"Cycle \d+: residual norm = (\S+)", run("pmg_main", *args, *extra))
        many = grab(r"Cycle \d+: residual norm = (\S+)", ranks(*extra))
        assert len(one) == len(many) == 4
        # no AMG: the same arithmetic; gathered set-up: the same hierarchy, sums ordered differently; distributed set-up
        # (the default on several ranks): another hierarchy -- aggregates stay inside a rank -- with the same contraction
        tol = 1e-9 if not extra else 1e-5 if "gathered" in extra else 1e-3
        assert all(abs(a - b) < tol * one[0] for a, b in zip(one, many)), (one, many)
    out = ranks("--graph", "--pcg")
    assert "(hipGraph replays)" in out
    m = re.search(r"PCG with V-cycle preconditioner: (\d+) iterations, \|b - A x\| / \|b\| = ([0-9.e+-]+)", out)
    assert m and float(m.group(2)) < 1e-6


def test_cg_and_vector_update_drivers_as_ranks_sharing_the_gpu(built):
    """cg_main and vector_update_main as two processes on the one GPU (--comm windows): the same global problem as the
    single-process run -- same CG iteration count and eigenvalue estimates, same norms."""
    script = os.path.join(ROOT, "examples", "pmg", "run_ranks.sh")

    def ranks(exe, *args):
        r = subprocess.run(["bash", script, "1,1,2", *map(str, args), "--comm", "windows"], capture_output=True,
                           text=True, timeout=300,
                           env={**os.environ, "PMG_MAIN": os.path.join(BIN, exe), "PMG_WINDOW_TIMEOUT_MS": "20000"})
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    one, two = run("cg_main", "--n", 6, "--degree", 3), ranks("cg_main", "--n", 6, "--degree", 3)
    assert grab(r"Number of iterations (\d+)", one) == grab(r"Number of iterations (\d+)", two)
    e1 = re.findall(r"Computed eigs = \(([0-9.e+-]+), ([0-9.e+-]+)\)", one)[0]
    e2 = re.findall(r"Computed eigs = \(([0-9.e+-]+), ([0-9.e+-]+)\)", two)[0]
    assert all(abs(float(a) - float(b)) < 1e-9 * float(e1[1]) for a, b in zip(e1, e2))
    (b1,), (b2,) = grab(r"Norm of b = (\S+)", one), grab(r"Norm of b = (\S+)", two)
    assert abs(b1 - b2) < 1e-11 * b1
    out = ranks("vector_update_main", "--n", 8, "--degree", 2, "--iterations", 20)
    assert "Ghost check" in out and len(grab(r"Dot value: (\S+)", out)) == 20
