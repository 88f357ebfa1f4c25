"""The multi-rank path on real kernels: two ranks share the one GPU of the test
box and exchange halos through a gloo process group (staged through host
memory -- RCCL needs one GPU per rank, which this box does not have; the RCCL
branch differs only in who moves the packed buffers).  Checks the brick
partition + ghost layer + pack/unpack kernels + interior/boundary patch
launches + distributed dot products against the single-domain CPU oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def warp(x):
    return x + 0.03 * np.sin(3.0 * x[:, [1, 2, 0]])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, dims, orders, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pmg_dolfinx_amd as pm
        from oracle import pmg_oracle as po

        torch.cuda.set_device(0)
        k = 3
        H = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, proc_dims=dims, rank=rank, size=world, warp=warp)
        out = {"eig": H.eig_ranges, "ghosts": [lv.num_ghosts for lv in H.levels]}
        # operator apply on every level against the global oracle
        gm = po.BoxMesh(n, warp=warp)
        errs = []
        for P, lv, layout, op in zip(orders, H.levels, H.layouts, H.operators):
            A = po.Laplacian(P, 2.0, gm.dofmap(P), gm.xgeom, gm.geom_dofmap, gm.boundary_marker(P))
            ug = np.random.default_rng(11).standard_normal(A.ndofs)
            x, y = pm.Vector(layout), pm.Vector(layout)
            xl = np.zeros(lv.ndofs)
            xl[: lv.size_local] = ug[lv.local_to_global[: lv.size_local]]  # ghosts stale: apply must update them
            x.data.copy_(torch.from_numpy(xl))
            op(x, y)
            ref = A.apply(ug)[lv.local_to_global[: lv.size_local]]
            errs.append(float(np.abs(y.data_copy()[: lv.size_local] - ref).max() / np.abs(ref).max()))
            # the ghosts of the input were refreshed as a side effect (src/laplacian.hpp:378,425)
            assert np.array_equal(x.data_copy(), ug[lv.local_to_global])
            # distributed reductions
            assert abs(pm.inner_product(x, x) - ug @ ug) < 1e-10 * (ug @ ug)
            assert pm.norm(x, "linf") == np.abs(ug).max()  # max-reduction over the ranks (src/vector.hpp:383-385)
        out["apply_err"] = errs
        # V-cycles against the single-domain oracle with the same smoother bounds
        mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=warp)
        for s, e in zip(sm, H.eig_ranges):
            s.eig_range = e
        lvf = H.levels[-1]
        xv = H.new_vector()
        xv.set(0.0)
        xo = np.zeros_like(b)
        verr = []
        for _ in range(2):
            rn = H.mg.apply(H.rhs[-1], xv, verbose=True)
            xo = mg.apply(b, xo, compute_rnorm=True)
            ref = xo[lvf.local_to_global[: lvf.size_local]]
            verr.append((float(np.abs(xv.data_copy()[: lvf.size_local] - ref).max() / np.abs(xo).max()),
                         abs(rn - mg.rnorm) / mg.rnorm))
        out["vcycle_err"] = verr
        out["eig_ref"] = eigs
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (3, 4, 8)), ((2, 1, 1), (6, 3, 4)), ((1, 2, 2), (3, 4, 6))])
def test_ranks_share_one_gpu(dims, n, built):
    """Two ranks (one neighbour each) and four ranks (three neighbours each, edge and corner
    exchanges) on the one GPU."""
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    orders = (1, 2, 4) if dims[0] * dims[1] * dims[2] == 2 else (1, 2)
    world = dims[0] * dims[1] * dims[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, dims, orders, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert all(g > 0 for g in out["ghosts"])
        assert max(out["apply_err"]) < 1e-12, out["apply_err"]
        for got, ref in zip(out["eig"], out["eig_ref"]):
            assert abs(got[1] - ref[1]) < 1e-8 * ref[1]
        for e, rn in out["vcycle_err"]:
            assert e < 1e-10 and rn < 1e-8


def _rccl_worker(port, q):
    """One rank, nccl (= RCCL) process group: the rank is its own neighbour, so the
    production exchange branch (async all_to_all_single on RCCL's stream, wait on the
    compute stream) and the device all-reduce run for real on the one GPU."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        import pmg_dolfinx_amd as pm

        rng = np.random.default_rng(5)
        n, m = 100_000, 30_000
        send = rng.permutation(n)[:m].astype(np.int32)
        layout = pm.Layout(n, m, neighbors=[0], send_counts=[m], recv_counts=[m], send_indices=send,
                           recv_indices=np.arange(m, dtype=np.int32), always_exchange=True)
        assert layout.distributed and not layout._staged
        out = {}
        side = torch.cuda.Stream()
        for name, stream in (("default", torch.cuda.current_stream()), ("side", side)):
            with torch.cuda.stream(stream):
                x = pm.Vector(layout)
                a = rng.standard_normal(n + m)
                x.data.copy_(torch.from_numpy(a))
                for _ in range(3):  # repeated begin/end pairs reuse the staging buffers
                    x.scatter_fwd_begin()
                    pm.scale(x, 1.0)  # work enqueued between begin and end, like the interior cells
                    x.scatter_fwd_end()
                got = x.data_copy()
                out[name + "_fwd"] = bool(np.array_equal(got[n:], a[send]) and np.array_equal(got[:n], a[:n]))
                # reverse: ghosts accumulate into their owners
                x.scatter_rev_begin()
                x.scatter_rev_end()
                ref = a[:n].copy()
                np.add.at(ref, send, a[send])
                out[name + "_rev"] = float(np.abs(x.data_copy()[:n] - ref).max())
                out[name + "_dot"] = abs(pm.inner_product(x, x) - ref @ ref) / (ref @ ref)
                out[name + "_linf"] = bool(pm.norm(x, "linf") == np.abs(x.data_copy()[:n]).max())
        # a hierarchy under an initialised RCCL group still runs (single brick: no exchange partners)
        H = pm.PoissonHierarchy(4, (1, 2), cheb_its=2)
        v = H.new_vector()
        v.set(0.0)
        out["rnorm"] = H.mg.apply(H.rhs[-1], v, verbose=True)
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_branch_single_rank(built):
    import torch
    import torch.multiprocessing as mp

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    for s in ("default", "side"):
        assert out[s + "_fwd"]
        assert out[s + "_rev"] < 1e-14 and out[s + "_dot"] < 1e-13 and out[s + "_linf"]
    assert np.isfinite(out["rnorm"]) and out["rnorm"] > 0
