"""Host logic of the multi-GPU path on CPU: brick partition + ghost layer
(src/mesh.hpp), halo index lists (Scatterer), and the exchange itself over
torch.distributed with the gloo backend at world_size 2."""
import os
import socket

import numpy as np
import pytest

from oracle import pmg_oracle as po


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("dims,n", [((2, 2, 2), (4, 5, 6)), ((1, 1, 2), (3, 3, 4)), ((3, 1, 2), (3, 2, 4)),
                                     ((1, 2, 4), (2, 4, 8))])
def test_partition_consistency(dims, n, built):
    import pmg_dolfinx_amd as pm

    size = int(np.prod(dims))
    parts = [pm.BoxPartition(n, dims, r) for r in range(size)]
    gm = po.BoxMesh(n)
    for P in (1, 2, 4):
        lvs = [q.level(P) for q in parts]
        ng = parts[0].global_ndofs(P)
        owned = np.concatenate([lv.local_to_global[: lv.size_local] for lv in lvs])
        assert len(owned) == ng and len(np.unique(owned)) == ng  # every dof owned exactly once
        for r, lv in enumerate(lvs):
            off = 0
            for q, cnt in zip(lv.neighbors, lv.recv_counts):
                lq = lvs[q]
                i = lq.neighbors.index(r)
                so = sum(lq.send_counts[:i])
                assert lq.send_counts[i] == cnt
                sent = lq.local_to_global[lq.send_indices[so: so + cnt]]
                assert np.array_equal(sent, lv.local_to_global[lv.size_local + off: lv.size_local + off + cnt])
                assert (lv.ghost_owners[off: off + cnt] == q).all()
                off += cnt
            assert off == lv.num_ghosts
            assert (lv.dofmap[lv.lcells] < lv.size_local).all()  # src/mesh.hpp:120-126
            assert set(lv.lcells) | set(lv.bcells) == set(range(parts[r].ncells))
        # ghost layer makes every owned operator row complete (src/mesh.hpp:11-12)
        A = po.Laplacian(P, 2.0, gm.dofmap(P), gm.xgeom, gm.geom_dofmap, gm.boundary_marker(P))
        u = np.random.default_rng(0).standard_normal(ng)
        y = A.apply(u)
        for q, lv in zip(parts, lvs):
            Al = po.Laplacian(P, 2.0, lv.dofmap, q.xgeom, q.geom_dofmap, lv.bc_marker)
            yl = Al.apply(u[lv.local_to_global])
            assert np.abs(yl[: lv.size_local] - y[lv.local_to_global[: lv.size_local]]).max() < 1e-12


def test_single_rank_matches_oracle_mesh(built):
    import pmg_dolfinx_amd as pm

    m, bp = po.BoxMesh((3, 4, 5)), pm.BoxPartition((3, 4, 5))
    for P in (1, 2, 3):
        lv = bp.level(P)
        assert np.array_equal(lv.dofmap, m.dofmap(P)) and lv.num_ghosts == 0
        assert np.array_equal(lv.bc_marker, m.boundary_marker(P))
        assert len(lv.bcells) == 0 and np.array_equal(lv.lcells, np.arange(60))
    assert np.allclose(bp.dof_coordinates(2), m.dof_coordinates(2))
    assert pm.default_proc_dims(8) == (2, 2, 2) and pm.default_proc_dims(2) == (1, 1, 2)
    assert pm.default_proc_dims(4) == (1, 2, 2) and pm.default_proc_dims(1) == (1, 1, 1)


def _gloo_worker(rank, world, port, n, dims, q):
    import torch
    import torch.distributed as dist

    import pmg_dolfinx_amd as pm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        part = pm.BoxPartition(n, dims, rank)
        gm = po.BoxMesh(n)
        out = {}
        for P in (1, 2, 4):
            lv = part.level(P)
            layout = pm.Layout(lv.size_local, lv.num_ghosts, lv.neighbors, lv.send_counts, lv.recv_counts,
                               lv.send_indices, lv.recv_indices, device="cpu")
            # 1. halo: owners' values reach every ghost copy
            x = np.full(lv.ndofs, -1.0)
            x[: lv.size_local] = lv.local_to_global[: lv.size_local].astype(np.float64)
            layout.scatter_fwd_host(x)
            ok_halo = np.array_equal(x, lv.local_to_global.astype(np.float64))
            # 2. operator apply on the local mesh after the exchange == global apply on owned rows
            A = po.Laplacian(P, 2.0, gm.dofmap(P), gm.xgeom, gm.geom_dofmap, gm.boundary_marker(P))
            ug = np.random.default_rng(3).standard_normal(A.ndofs)
            u = np.zeros(lv.ndofs)
            u[: lv.size_local] = ug[lv.local_to_global[: lv.size_local]]
            layout.scatter_fwd_host(u)
            Al = po.Laplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
            yl = Al.apply(u)[: lv.size_local]
            err = float(np.abs(yl - A.apply(ug)[lv.local_to_global[: lv.size_local]]).max())
            # 3. distributed dot over owned entries == global dot
            t = torch.tensor([float(u[: lv.size_local] @ u[: lv.size_local])], dtype=torch.float64)
            dist.all_reduce(t)
            out[P] = (ok_halo, err, abs(t.item() - float(ug @ ug)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (3, 3, 4)), ((2, 1, 1), (4, 2, 3))])
def test_halo_exchange_gloo_world2(dims, n, built):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n, dims, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        for P, (ok_halo, err, derr) in out.items():
            assert ok_halo, (rank, P)
            assert err < 1e-12 and derr < 1e-9


def test_cpp_brick_partition_is_consistent(built):
    """examples/common/brick_partition.hpp (the C++ drivers' partitioner, same rules as mesh.py): every
    global dof owned once, both sides of every halo list agree entry by entry -- host only."""
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pmg-dolfinx_amd", "bin", "pmg_main")
    for n, dims, orders in ((6, "2,2,2", "1,2,4"), (7, "2,3,2", "1,3"), (5, "1,1,2", "2"), (8, "2,1,4", "1")):
        r = subprocess.run([exe, "--n", str(n), "--orders", orders, "--check-partition", dims], capture_output=True,
                           text=True, timeout=120)
        assert r.returncode == 0 and "consistent" in r.stdout, r.stdout + r.stderr
    # and against the Python partitioner: same sizes on every rank
    from pmg_dolfinx_amd.mesh import BoxPartition
    for rank in range(8):
        lv = BoxPartition(6, (2, 2, 2), rank).level(2)
        assert lv.size_local > 0 and len(lv.neighbors) == 7
