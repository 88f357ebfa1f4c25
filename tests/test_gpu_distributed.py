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


def _run_ranks(target, world, args, timeout=300):
    """Spawn `world` processes running target(rank, world, port, *args, q); every worker puts
    (rank, result) or (rank, exception text) on the queue.  Whatever happens -- a worker that dies
    before reporting, a failure reported by one rank while the others sit in a collective -- every
    process is terminated and joined before the test returns, so no rank is left holding the GPU."""
    import queue as _queue

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        while len(res) < world:
            try:
                rank, out = q.get(timeout=5)
            except _queue.Empty:
                timeout -= 5
                dead = [i for i, p in enumerate(procs) if p.exitcode not in (None, 0) and i not in res]
                if dead:
                    raise AssertionError(f"rank(s) {dead} died without reporting (exit codes "
                                         f"{[procs[i].exitcode for i in dead]})")
                if timeout <= 0:
                    raise AssertionError("timed out waiting for the ranks")
                continue
            if isinstance(out, str):
                raise AssertionError(f"rank {rank} failed:\n{out}")
            res[rank] = out
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=30)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [res[r] for r in range(world)]


def _reporting(fn):
    """Worker wrapper: report the traceback instead of dying silently."""
    import functools

    @functools.wraps(fn)
    def wrapped(rank, world, port, *args):
        q = args[-1]
        try:
            q.put((rank, fn(rank, world, port, *args[:-1])))
        except BaseException:  # noqa: BLE001 -- reported to the parent, which fails the test
            import traceback

            q.put((rank, traceback.format_exc()))
            raise

    return wrapped


def _rank_checks(pm, rank, world, n, dims, orders, comm=None):
    """What every rank checks against the single-domain oracle: operator apply on every level (with
    stale ghosts on input), distributed reductions, eigenvalue estimates, two V-cycles."""
    import torch

    from oracle import pmg_oracle as po

    k = 3
    H = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, proc_dims=dims, rank=rank, size=world, warp=warp,
                            comm=comm)
    out = {"eig": H.eig_ranges, "ghosts": [lv.num_ghosts for lv in H.levels],
           "neighbors": [len(lv.neighbors) for lv in H.levels]}
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
    # exchanges of one cycle (round 4): an operator application, a prolongation and a restriction each refresh the
    # ghosts of their input -- except the first application of every post-smooth, whose input u + P u_c has current
    # ghosts by construction (solvers.hip, local_correction).  Three levels, k = 3: 7 + 6 + 2 + 4 - 2 = 17.
    H.mg.set_graph(False)  # a replayed cycle issues its scatters once, at capture
    before = sum(l.forward_scatters() for l in H.layouts)
    H.mg.apply(H.rhs[-1], xv)
    H.mg.set_graph(None)
    L = len(orders)
    expect = (2 * k + 1) + 2 * k * (L - 2) + (k - 1) + 2 * (L - 1) - (L - 1) if L > 1 else k
    out["exchanges_per_cycle"] = (sum(l.forward_scatters() for l in H.layouts) - before, expect)
    # the coarsest level solved by CG + AMG (each rank's hierarchy on its own block: a block preconditioner
    # for the distributed Krylov solve) against the oracle's cycle with an exact coarse solve
    import scipy.sparse.linalg as spla

    lu = spla.splu(ops[0].assemble_csr().tocsc())

    def exact(u0, b0):
        u0[:] = lu.solve(b0)

    mgo = po.MultigridPreconditioner(ops, sm, it, mesh.boundary_marker(orders[0]), coarse_solver=exact)
    amg = pm.AmgSolver(H.operators[0], max_iter=100, rtol=1e-11)
    H.mg.set_coarse_solver(amg)
    xv.set(0.0)
    xo = np.zeros_like(b)
    aerr = []
    for _ in range(2):
        H.mg.apply(H.rhs[-1], xv)
        xo = mgo.apply(b, xo)
        ref = xo[lvf.local_to_global[: lvf.size_local]]
        aerr.append(float(np.abs(xv.data_copy()[: lvf.size_local] - ref).max() / np.abs(xo).max()))
    out["amg_vcycle_err"] = aerr
    # the replicated hierarchy (global coarse matrix on every rank, one all-reduce per solve): stationary
    # cycles work on several ranks, and a Krylov solve takes exactly as many iterations as on one rank
    lv0 = H.levels[0]
    g = np.random.default_rng(5).standard_normal(H.part.global_ndofs(orders[0]))
    g[mesh.boundary_marker(orders[0]).astype(bool)] = 0.0
    bl = pm.Vector(H.layouts[0])
    bl.data.copy_(torch.from_numpy(g[lv0.local_to_global]))
    rep = pm.AmgSolver(H.operators[0], max_iter=60, rtol=1e-9, global_index=lv0.local_to_global,
                       n_global=H.part.global_ndofs(orders[0]))
    xl = pm.Vector(H.layouts[0])
    out["rep_its"] = rep.solve(xl, bl)
    A0 = ops[0].assemble_csr()
    xs = spla.spsolve(A0.tocsc(), g)
    out["rep_err"] = float(np.abs(xl.data_copy()[: lv0.size_local] - xs[lv0.local_to_global[: lv0.size_local]]).max()
                           / np.abs(xs).max())
    out["rep_levels"] = rep.info()
    # the same hierarchy with level 0 replicated as well (round 2's form): same solution, same iteration count +- 1
    rep_full = pm.AmgSolver(H.operators[0], max_iter=60, rtol=1e-9, global_index=lv0.local_to_global,
                            n_global=H.part.global_ndofs(orders[0]), distributed_fine_level=False)
    xf = pm.Vector(H.layouts[0])
    out["rep_full_its"] = rep_full.solve(xf, bl)
    out["rep_dist_vs_full"] = float(np.abs(xf.data_copy()[: lv0.size_local] - xl.data_copy()[: lv0.size_local]).max()
                                    / np.abs(xs).max())
    del rep_full, xf
    # the set-up WITHOUT the global level-0 matrix (round 4): first coarsening per rank, only level 1 gathered.  The
    # hierarchy depends on the partition (aggregates stay inside a rank), so: the same solution, an iteration count
    # within one of the gathered hierarchy's, the same hierarchy on every rank below level 0, and no level of it as
    # large as the global degree-1 problem
    dis = pm.AmgSolver(H.operators[0], max_iter=60, rtol=1e-9, global_index=lv0.local_to_global,
                       n_global=H.part.global_ndofs(orders[0]), setup="distributed")
    xd = pm.Vector(H.layouts[0])
    out["dis_its"] = dis.solve(xd, bl)
    out["dis_err"] = float(np.abs(xd.data_copy()[: lv0.size_local] - xs[lv0.local_to_global[: lv0.size_local]]).max()
                           / np.abs(xs).max())
    out["dis_levels"] = dis.info()
    out["n_global0"] = H.part.global_ndofs(orders[0])
    with pytest.raises(RuntimeError, match="without gathering"):
        pm._lib.call("pmg_amg_set_distributed_fine_level", dis.handle, 0)
    del dis, xd
    reps = pm.AmgSolver(H.operators[0], cycles=8, global_index=lv0.local_to_global,
                        n_global=H.part.global_ndofs(orders[0]))
    H.mg.set_coarse_solver(reps)
    xv.set(0.0)
    xo = np.zeros_like(b)
    serr = []
    for _ in range(2):
        H.mg.apply(H.rhs[-1], xv)
        xo = mgo.apply(b, xo)
        ref = xo[lvf.local_to_global[: lvf.size_local]]
        serr.append(float(np.abs(xv.data_copy()[: lvf.size_local] - ref).max() / np.abs(xo).max()))
    out["rep_vcycle_err"] = serr
    H.mg.set_coarse_solver(None)
    return out


def _assert_rank_results(res):
    for out in res:
        assert all(g > 0 for g in out["ghosts"])
        assert max(out["apply_err"]) < 1e-12, out["apply_err"]
        for got, ref in zip(out["eig"], out["eig_ref"]):
            assert abs(got[1] - ref[1]) < 1e-8 * ref[1]
        for e, rn in out["vcycle_err"]:
            assert e < 1e-10 and rn < 1e-8
        assert out["exchanges_per_cycle"][0] == out["exchanges_per_cycle"][1], out["exchanges_per_cycle"]
        assert max(out["amg_vcycle_err"]) < 1e-7, out["amg_vcycle_err"]
        assert out["rep_err"] < 1e-7 and max(out["rep_vcycle_err"]) < 1e-5, (out["rep_err"], out["rep_vcycle_err"])
    # the replicated hierarchy is the same on every rank, and it is the single-rank hierarchy
    assert all(out["rep_its"] == res[0]["rep_its"] and out["rep_levels"] == res[0]["rep_levels"] for out in res)
    assert res[0]["rep_its"] <= 14
    for out in res:
        assert abs(out["rep_full_its"] - out["rep_its"]) <= 1 and out["rep_dist_vs_full"] < 1e-7, (
            out["rep_full_its"], out["rep_its"], out["rep_dist_vs_full"])
        # distributed set-up: the same solution (these meshes are so small that the gathered hierarchy is a direct
        # solve; iteration counts are compared on a mesh that coarsens, test_distributed_amg_setup_on_eight_ranks)
        assert out["dis_its"] <= 14 and out["dis_err"] < 1e-7, (out["dis_its"], out["rep_its"], out["dis_err"])
        assert out["dis_levels"][0]["rows"] < out["n_global0"]  # level 0 = the rank's own rows
        assert all(lv["rows"] < 0.5 * out["n_global0"] for lv in out["dis_levels"][1:])
    assert all(out["dis_levels"][1:] == res[0]["dis_levels"][1:] for out in res)


def _worker_body(rank, world, port, n, dims, orders, halo="exchange"):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("PMG_WINDOW_TIMEOUT_MS", "20000")  # the ranks time-slice one GPU: generous, still bounded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pmg_dolfinx_amd as pm

        torch.cuda.set_device(0)
        if halo == "windows-split":  # small levels as large ones: put, interior cells, get, boundary cells
            os.environ["PMG_FUSED_EXCHANGE"] = "0"
            halo = "windows"
        comm = {"windows": lambda: pm.TorchComm(halo="windows"),       # halo windows, reductions on the gloo callbacks
                "window-comm": lambda: pm.WindowComm.from_torch(),     # halo AND reductions through windows
                "exchange": lambda: None}[halo]()
        out = _rank_checks(pm, rank, world, n, dims, orders, comm=comm)
        if halo == "window-comm":
            # collectives longer than one exchange of the window (16 384 doubles per rank): chunked all-reduce and gather
            import ctypes as C

            from pmg_dolfinx_amd import _lib

            m = 40_000
            v = torch.arange(m, dtype=torch.float64, device="cuda") * (rank + 1)
            _lib.call("pmg_comm_allreduce_sum", comm.native, _lib.ptr(v), m, _lib.current_stream())
            torch.cuda.synchronize()
            want = np.arange(m, dtype=np.float64) * (world * (world + 1) // 2)
            out["long_allreduce_ok"] = bool(np.array_equal(v.cpu().numpy(), want))
            nbytes = 200_003
            mine = np.full(nbytes, rank + 1, dtype=np.uint8)
            mine[::1000] = np.arange(len(mine[::1000]), dtype=np.uint8)
            got = np.zeros(nbytes * world, dtype=np.uint8)
            _lib.call("pmg_comm_allgather", comm.native, _lib.vp(mine.ctypes.data), nbytes, _lib.vp(got.ctypes.data))
            ok = True
            for r in range(world):
                ref = np.full(nbytes, r + 1, dtype=np.uint8)
                ref[::1000] = np.arange(len(ref[::1000]), dtype=np.uint8)
                ok = ok and np.array_equal(got[r * nbytes:(r + 1) * nbytes], ref)
            out["long_allgather_ok"] = bool(ok)
        if halo != "exchange":
            # the same cycles replayed as a hipGraph: no host in the loop, so the ranks drift apart as far as the
            # protocol lets them (two exchanges) -- the exchange numbers kept on the device and the "consumed" counters
            # are what keeps the replays correct
            H = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=3, proc_dims=dims, rank=rank, size=world, warp=warp,
                                    comm=comm)

            def cycles(graph):
                H.mg.set_graph(graph)
                x = H.new_vector()
                x.set(0.0)
                for _ in range(8):
                    H.mg.apply(H.rhs[-1], x)
                torch.cuda.synchronize()
                return x.data_copy()[: H.levels[-1].size_local].copy(), H.mg.graph_replays()

            xe, r0 = cycles(False)
            xg, r1 = cycles(True)
            H.mg.set_graph(False)
            out["graph_replays"] = r1 - r0
            out["graph_vs_eager"] = float(np.abs(xg - xe).max() / np.abs(xe).max())
        dist.barrier()  # nobody frees a window a neighbour may still acknowledge into
        return out
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, n, dims, orders, *rest):
    _reporting(_worker_body)(rank, world, port, n, dims, orders, *rest)


@pytest.mark.parametrize("dims,n", [((1, 1, 2), (3, 4, 8)), ((2, 1, 1), (6, 3, 4)), ((1, 2, 2), (3, 4, 6))])
def test_ranks_share_one_gpu(dims, n, built):
    """Two ranks (one neighbour each) and four ranks (three neighbours each, edge and corner
    exchanges) on the one GPU."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    orders = (1, 2, 4) if dims[0] * dims[1] * dims[2] == 2 else (1, 2)
    world = dims[0] * dims[1] * dims[2]
    _assert_rank_results(_run_ranks(_worker, world, (n, dims, orders)))


@pytest.mark.parametrize("route", ["windows", "window-comm", "windows-split"])
@pytest.mark.parametrize("dims,n", [((1, 1, 2), (3, 4, 8)), ((1, 2, 2), (3, 4, 6)),
                                    ((1, 1, 3), (3, 4, 9))])  # a chain: the middle rank has two neighbours, the ends
                                                               # one -- a rank's place in its neighbours' lists differs
def test_ranks_share_one_gpu_through_halo_windows(dims, n, route, built):
    """The same checks with the halo moved by the library's windows: every rank (a process) stores its packed
    values straight into its neighbours' interprocess-mapped windows and waits on their flags -- the whole protocol of
    window.hip between real processes (two and four: the box admits six on its GPU, the test runner included), with the one GPU standing in for the
    peers' GPUs.  route "windows": the reductions stay on the gloo callbacks; "window-comm": the library's communicator
    made of windows -- no transport library at all, every reduction (the dot products of CG, the sums of whole level
    vectors in the replicated coarse solve: chunked) two kernels of direct stores and flags; "windows-split": the levels
    of these small meshes take their exchange whole, in one launch in front of one launch over all cells (round 4,
    laplacian_single_launch) -- this route runs them the way a large level runs, PMG_FUSED_EXCHANGE=0.  Not run with the ranks as threads of one process: eight
    streams share the process's four hardware queues, and a kernel waiting for a flag at the head of a queue would
    hold back the very kernel that raises it."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    orders = (1, 2, 4) if dims[0] * dims[1] * dims[2] == 2 else (1, 2)
    world = dims[0] * dims[1] * dims[2]
    res = _run_ranks(_worker, world, (n, dims, orders, route))
    _assert_rank_results(res)
    for out in res:
        assert out["graph_replays"] >= 7 and out["graph_vs_eager"] < 1e-12, (out["graph_replays"], out["graph_vs_eager"])
        if route == "window-comm":
            assert out["long_allreduce_ok"] and out["long_allgather_ok"]


# ---------------------------------------------------------------------------------------------
# BASELINE config 3's partition -- 2 x 2 x 2 bricks, every rank with 7 neighbours (3 faces, 3 edges,
# 1 corner) -- on the one GPU of the test box: the eight ranks are eight host THREADS of one process
# (the box admits at most 6 processes on its GPU), each with its own HIP stream and its own library
# objects; the exchange callback of the C ABI moves the packed buffers between the ranks' staging
# buffers with stream-ordered device copies.
class _ThreadWorld:
    def __init__(self, n):
        import threading

        self.n = n
        self.barrier = threading.Barrier(n)
        self.layouts = [dict() for _ in range(n)]
        self.slots = [None] * n

    def wait(self):
        self.barrier.wait(timeout=240)


class _ThreadComm:
    native = None
    staged = False
    distributed = True

    def __init__(self, world, rank):
        self.W, self.rank, self.world = world, rank, world.n
        self._count = 0

    def register(self, layout):  # layouts are created in the same order on every rank
        layout._tid = self._count
        self.W.layouts[self.rank][self._count] = layout
        self._count += 1

    def exchange(self, L, phase):
        import torch

        st = torch.cuda.current_stream()
        W = self.W
        if phase in (0, 2):
            fwd = phase == 0
            L._ev_packed = torch.cuda.Event()
            L._ev_packed.record(st)
            W.wait()  # every rank has packed (on its own stream) and published its event
            off = 0
            for i, nb in enumerate(L.neighbors):
                peer = W.layouts[nb][L._tid]
                cnt = (L.recv_counts if fwd else L.send_counts)[i]
                j = peer.neighbors.index(self.rank)
                pc = peer.send_counts if fwd else peer.recv_counts
                assert pc[j] == cnt, "the two sides of a halo plan disagree"
                po_ = sum(pc[:j])
                st.wait_event(peer._ev_packed)
                src = (peer.send_buffer if fwd else peer.recv_buffer)[po_: po_ + cnt]
                dst = (L.recv_buffer if fwd else L.send_buffer)[off: off + cnt]
                dst.copy_(src)
                off += cnt
            L._ev_copied = torch.cuda.Event()
            L._ev_copied.record(st)
        else:
            W.wait()  # every rank has enqueued its copies: nobody repacks a buffer that is still being read
            for nb in L.neighbors:
                st.wait_event(W.layouts[nb][L._tid]._ev_copied)

    def allreduce(self, L, host, op):
        W = self.W
        W.slots[self.rank] = host.copy()
        W.wait()
        vals = np.stack(W.slots)
        tot = vals.max(axis=0) if op == "max" else vals.sum(axis=0)
        W.wait()
        host[:] = tot


def test_distributed_amg_setup_on_eight_ranks(built):
    """the AMG set up WITHOUT the global degree-1 matrix -- every rank coarsens its own block with one
    layer of overlap, only level 1 is gathered -- on the 2 x 2 x 2 split (eight ranks as threads), 24^3 cells = 15 625
    degree-1 dofs: the same solution; iteration counts within one of the single-rank hierarchy's (the gathered hierarchy
    IS that hierarchy; the distributed one depends on the partition -- aggregates do not cross rank boundaries; the
    rank-local block preconditioner of pmg_amg_create takes 40, tools/amg_rank_scaling.py); no rank holds a level as
    large as the global problem; set-up times printed."""
    import threading
    import time

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import scipy.sparse.linalg as spla

    import pmg_dolfinx_amd as pm
    from oracle import pmg_oracle as po

    n, dims, world = (24, 24, 24), (2, 2, 2), 8
    gm = po.BoxMesh(n, warp=warp)
    A = po.Laplacian(1, 2.0, gm.dofmap(1), gm.xgeom, gm.geom_dofmap, gm.boundary_marker(1))
    g = np.random.default_rng(5).standard_normal(A.ndofs)
    g[gm.boundary_marker(1).astype(bool)] = 0.0
    xs = spla.spsolve(A.assemble_csr().tocsc(), g)
    # one rank, the whole mesh: the reference iteration count
    part1 = pm.BoxPartition(n, warp=warp)
    lv1 = part1.level(1)
    lay1 = pm.make_layout(lv1)
    op1 = pm.MatFreeLaplacian(1, 2.0, lv1.dofmap, part1.xgeom, part1.geom_dofmap, lv1.lcells, lv1.bcells,
                              lv1.bc_marker, lay1)
    one = pm.AmgSolver(op1, max_iter=60, rtol=1e-9)
    b1, x1 = pm.Vector(lay1), pm.Vector(lay1)
    b1.data.copy_(torch.from_numpy(g))
    its_one = one.solve(x1, b1)
    assert np.abs(x1.data_copy() - xs).max() < 1e-7 * np.abs(xs).max()
    del one, op1, b1, x1

    W = _ThreadWorld(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                comm = _ThreadComm(W, rank)
                part = pm.BoxPartition(n, dims, rank, warp=warp)
                lv = part.level(1)
                layout = pm.make_layout(lv, comm=comm)
                op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                                         lv.bc_marker, layout)
                bl = pm.Vector(layout)
                bl.data.copy_(torch.from_numpy(g[lv.local_to_global]))
                out = {}
                for setup in ("gathered", "distributed"):
                    t0 = time.perf_counter()
                    amg = pm.AmgSolver(op, max_iter=60, rtol=1e-9, global_index=lv.local_to_global, n_global=A.ndofs,
                                       setup=setup)
                    torch.cuda.current_stream().synchronize()
                    out[setup + "_setup_s"] = time.perf_counter() - t0
                    xl = pm.Vector(layout)
                    out[setup + "_its"] = amg.solve(xl, bl)
                    out[setup + "_err"] = float(np.abs(xl.data_copy()[: lv.size_local]
                                                       - xs[lv.local_to_global[: lv.size_local]]).max() / np.abs(xs).max())
                    out[setup + "_levels"] = amg.info()
                    del amg, xl
                res[rank] = out
                torch.cuda.current_stream().synchronize()
        except BaseException:  # noqa: BLE001
            import traceback

            errors.append((rank, traceback.format_exc()))
            W.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, "\n".join(f"rank {r}:\n{tb}" for r, tb in errors)
    print([(o["gathered_its"], o["gathered_err"], o["distributed_its"], o["distributed_err"]) for o in res])
    for out in res:
        assert out["gathered_err"] < 1e-7 and out["distributed_err"] < 1e-7
        assert abs(out["gathered_its"] - its_one) <= 1, (out["gathered_its"], its_one)
        assert abs(out["distributed_its"] - its_one) <= 1, (out["distributed_its"], its_one)
        lv_d = out["distributed_levels"]
        assert lv_d[0]["rows"] < 0.2 * A.ndofs                      # level 0: the rank's own rows only
        assert all(l["rows"] < 0.25 * A.ndofs for l in lv_d[1:])    # nothing gathered is as large as level 0
        assert out["gathered_levels"][0]["rows"] == A.ndofs         # ... which the gathered set-up holds on every rank
    assert all(out["distributed_levels"][1:] == res[0]["distributed_levels"][1:] for out in res)
    print("AMG set-up on 8 ranks (threads of one process, one GPU): gathered "
          f"{max(o['gathered_setup_s'] for o in res):.2f} s, distributed {max(o['distributed_setup_s'] for o in res):.2f} s; "
          f"iterations one rank / gathered / distributed: {its_one} / {res[0]['gathered_its']} / {res[0]['distributed_its']}")


@pytest.mark.parametrize("n,orders", [((8, 8, 8), (1, 2, 4)),   # BASELINE config 3's levels, 4 x 4 x 4 cells per brick
                                      ((6, 6, 6), (1, 3, 6))])  # BASELINE config 5's: p = 6 -> 3 -> 1 + AMG at p = 1
def test_eight_ranks_2x2x2_as_threads(built, n, orders):
    import threading

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pmg_dolfinx_amd as pm

    dims = (2, 2, 2)
    world = 8
    W = _ThreadWorld(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                res[rank] = _rank_checks(pm, rank, world, n, dims, orders, comm=_ThreadComm(W, rank))
                torch.cuda.current_stream().synchronize()
        except BaseException:  # noqa: BLE001
            import traceback

            errors.append((rank, traceback.format_exc()))
            W.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, "\n".join(f"rank {r}:\n{tb}" for r, tb in errors)
    assert all(r is not None for r in res)
    assert all(nb == 7 for out in res for nb in out["neighbors"])  # faces, edges and the corner
    _assert_rank_results(res)


def _rccl_worker_body(rank, world, port):
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
        return out
    finally:
        dist.destroy_process_group()


def _rccl_worker(rank, world, port, q):
    _reporting(_rccl_worker_body)(rank, world, port, q)


def _native_worker_body(rank, world, port, halo="exchange"):
    """The library's own communicator (pmg_comm: RCCL bound at run time, grouped ncclSend/ncclRecv on
    its own stream, ncclAllReduce on device scalars) with one rank that is its own neighbour -- RCCL
    needs one GPU per rank, so on this box that is the whole of the native path that can run; the
    plan logic it shares with the other transports is covered by the multi-rank tests above."""
    import torch

    import pmg_dolfinx_amd as pm
    from oracle import pmg_oracle as po

    torch.cuda.set_device(0)
    comm = pm.RcclComm(0, 1, pm.RcclComm.unique_id(), halo=halo)
    rng = np.random.default_rng(6)
    n, m = 120_000, 40_000
    send = rng.permutation(n)[:m].astype(np.int32)
    # two "neighbours", both this rank, with different counts: segment offsets matter
    m0 = 15_000
    layout = pm.Layout(n, m, neighbors=[0, 0], send_counts=[m0, m - m0], recv_counts=[m0, m - m0], send_indices=send,
                       recv_indices=np.arange(m, dtype=np.int32), comm=comm)
    out = {}
    side = torch.cuda.Stream()
    for name, stream in (("default", torch.cuda.current_stream()), ("side", side)):
        with torch.cuda.stream(stream):
            x = pm.Vector(layout)
            a = rng.standard_normal(n + m)
            x.data.copy_(torch.from_numpy(a))
            for _ in range(3):
                x.scatter_fwd_begin()
                pm.scale(x, 1.0)  # work between begin and end, like the interior cells
                x.scatter_fwd_end()
            got = x.data_copy()
            out[name + "_fwd"] = bool(np.array_equal(got[n:], a[send]) and np.array_equal(got[:n], a[:n]))
            x.scatter_rev_begin()
            x.scatter_rev_end()
            ref = a[:n].copy()
            np.add.at(ref, send, a[send])
            out[name + "_rev"] = float(np.abs(x.data_copy()[:n] - ref).max())
            out[name + "_dot"] = abs(pm.inner_product(x, x) - ref @ ref) / (ref @ ref)
            out[name + "_linf"] = bool(pm.norm(x, "linf") == np.abs(x.data_copy()[:n]).max())
    # the solvers on a layout with a communicator: reductions go through ncclAllReduce, the CG keeps its
    # scalars on the device; same iterates as the oracle
    orders, k, nmesh = (1, 2, 4), 3, 4
    H = pm.PoissonHierarchy(nmesh, orders, cheb_its=k, warp=warp, comm=comm)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(nmesh, orders, cheb_its=k, warp=warp)
    out["eig_err"] = max(abs(g[1] - e[1]) / e[1] for g, e in zip(H.eig_ranges, eigs))
    for s_, e in zip(sm, H.eig_ranges):
        s_.eig_range = e
    v = H.new_vector()
    v.set(0.0)
    rn = H.mg.apply(H.rhs[-1], v, verbose=True)
    xo = mg.apply(b, np.zeros_like(b), compute_rnorm=True)
    out["vcycle_err"] = float(np.abs(v.data_copy() - xo).max() / np.abs(xo).max())
    out["rnorm_err"] = abs(rn - mg.rnorm) / mg.rnorm
    cg = pm.CGSolver(H.layouts[-1])
    cg.set_max_iterations(30)
    cg.set_tolerance(1e-9)
    xs = H.new_vector()
    xs.set(0.0)
    out["pcg_its"] = cg.solve(H.operators[-1], xs, H.rhs[-1], preconditioner=H.mg)
    ocg = po.CGSolver()
    ocg.set_max_iterations(30)
    ocg.set_tolerance(1e-9)
    xr = np.zeros_like(b)
    out["pcg_its_ref"] = ocg.solve(ops[-1], xr, b, precond=lambda r: mg.apply(r, np.zeros_like(r)))
    out["pcg_err"] = float(np.abs(xs.data_copy() - xr).max() / np.abs(xr).max())
    return out


def _native_worker(rank, world, port, *rest):
    _reporting(_native_worker_body)(rank, world, port, *rest)


@pytest.mark.parametrize("halo", ["exchange", "windows"])
def test_native_rccl_communicator_single_rank(built, halo):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (out,) = _run_ranks(_native_worker, 1, (halo,))
    for s in ("default", "side"):
        assert out[s + "_fwd"]
        assert out[s + "_rev"] < 1e-14 and out[s + "_dot"] < 1e-13 and out[s + "_linf"]
    assert out["eig_err"] < 1e-8 and out["vcycle_err"] < 1e-10 and out["rnorm_err"] < 1e-8
    assert out["pcg_its"] == out["pcg_its_ref"] and out["pcg_err"] < 1e-8


def test_rccl_branch_single_rank(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (out,) = _run_ranks(_rccl_worker, 1, ())
    for s in ("default", "side"):
        assert out[s + "_fwd"]
        assert out[s + "_rev"] < 1e-14 and out[s + "_dot"] < 1e-13 and out[s + "_linf"]
    assert np.isfinite(out["rnorm"]) and out["rnorm"] > 0


def _self_partner_graph_body(rank, world, port, halo="exchange"):
    """A rank that is its own halo partner: every scatter packs, sends to itself through the library's
    communicator and unpacks real halo volumes.  The numbers are not the multi-rank solution, but they
    are a deterministic function of what the exchange moves -- so the same cycles replayed as a hipGraph
    (exchange captured on the compute stream) must reproduce the eager ones."""
    import torch

    import pmg_dolfinx_amd as pm
    from pmg_dolfinx_amd import problem

    torch.cuda.set_device(0)
    native = pm.RcclComm(0, 1, pm.RcclComm.unique_id(), halo=halo)
    orig = problem.make_layout

    def self_layout(comm, reverse=False):
        # reverse: the same (owned dof -> ghost) pairs listed backwards -- the same exchange through different
        # pack / unpack lists
        def make(lv, group=None, device="cuda", comm_=None, **kw):
            m = min(sum(lv.send_counts), sum(lv.recv_counts))
            si, ri = np.asarray(lv.send_indices[:m]), np.asarray(lv.recv_indices[:m])
            if reverse:
                si, ri = si[::-1].copy(), ri[::-1].copy()
            return pm.Layout(lv.size_local, lv.num_ghosts, [0] if m else [], [m] if m else [], [m] if m else [],
                             si, ri, device=device, comm=comm)
        return lambda lv, group=None, device="cuda", comm=None: make(lv, group, device)

    def cycles(H, graph):
        H.mg.set_graph(graph)
        x = H.new_vector()
        x.set(0.0)
        for _ in range(4):
            H.mg.apply(H.rhs[-1], x)
        torch.cuda.synchronize()
        return x.data_copy()[: H.levels[-1].size_local].copy(), H.mg.graph_replays()

    out = {}
    try:
        problem.make_layout = self_layout(native)
        H = pm.PoissonHierarchy((4, 4, 8), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)
        out["ghosts"] = [lv.num_ghosts for lv in H.levels]
        xe, r0 = cycles(H, False)
        xg, r1 = cycles(H, True)
        out["replays"] = r1 - r0
        out["graph_vs_eager"] = float(np.abs(xg - xe).max() / np.abs(xe).max())
        # ... and with the replicated AMG coarse solve, whose all-reduce is captured the same way
        lv0 = H.levels[0]
        # (one rank, so the "global" numbering is the owned one; the ghost dofs, owned by nobody here, are folded
        # onto owned ones -- a glued mesh, still a symmetric positive definite coarse matrix)
        gi = np.concatenate([np.arange(lv0.size_local), np.arange(lv0.num_ghosts) % lv0.size_local]).astype(np.int64)
        amg = pm.AmgSolver(H.operators[0], global_index=gi, n_global=lv0.size_local, cycles=2)
        H.mg.set_coarse_solver(amg)
        xa, r2 = cycles(H, False)
        xb, r3 = cycles(H, True)
        out["amg_replays"] = r3 - r2
        out["amg_graph_vs_eager"] = float(np.abs(xb - xa).max() / np.abs(xa).max())
        H.mg.set_coarse_solver(None)
        del amg, H
        # the same halo with its index lists in another order
        problem.make_layout = self_layout(native, reverse=True)
        H = pm.PoissonHierarchy((4, 4, 8), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)
        xs, _ = cycles(H, False)
        out["reordered_lists"] = float(np.abs(xs - xe).max() / np.abs(xe).max())
        del H
        problem.make_layout = lambda lv, group=None, device="cuda", comm=None: pm.Layout(
            lv.size_local, lv.num_ghosts, device=device)
        H = pm.PoissonHierarchy((4, 4, 8), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)
        xn, _ = cycles(H, False)
        out["exchange_matters"] = float(np.abs(xn - xe).max() / np.abs(xe).max())
    finally:
        problem.make_layout = orig
    return out


def _self_partner_graph_worker(rank, world, port, *rest):
    _reporting(_self_partner_graph_body)(rank, world, port, *rest)


@pytest.mark.parametrize("halo", ["exchange", "windows"])
def test_graph_replay_captures_the_rccl_exchange(built, halo):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (out,) = _run_ranks(_self_partner_graph_worker, 1, (halo,))
    assert all(g > 0 for g in out["ghosts"])
    assert out["replays"] >= 3  # captured on the first cycle, replayed afterwards
    assert out["graph_vs_eager"] < 1e-12  # tolerance: atomic-order noise of the merged launches
    assert out["exchange_matters"] > 1e-6  # the check is sensitive to what the exchange moves
    assert out["amg_replays"] >= 3 and out["amg_graph_vs_eager"] < 1e-12
    assert out["reordered_lists"] < 1e-12


def _segments_body(rank, world, port, halo="exchange"):
    """Several neighbour segments (here: three, all to the rank itself) through the library's communicator: the
    per-neighbour offsets of the padded staging buffers, forward and reverse."""
    import torch

    import pmg_dolfinx_amd as pm

    torch.cuda.set_device(0)
    comm = pm.RcclComm(0, 1, pm.RcclComm.unique_id(), halo=halo)
    n_local, counts = 5000, [37, 1001, 64]  # odd lengths: every later segment would start unaligned without padding
    m = sum(counts)
    rng = np.random.default_rng(3)
    send = rng.permutation(n_local)[:m].astype(np.int32)  # distinct owned dofs
    recv = rng.permutation(m).astype(np.int32)            # ghosts in a scrambled order
    lay = pm.Layout(n_local, m + 5, [0, 0, 0], counts, counts, send, recv, comm=comm)
    x = pm.Vector(lay)
    x0 = rng.standard_normal(n_local + m + 5)
    x.data.copy_(torch.as_tensor(x0, device="cuda"))
    x.scatter_fwd()
    torch.cuda.synchronize()
    xf = x.data_copy()
    ref = x0.copy()
    ref[n_local + recv] = x0[send]
    out = {"fwd": float(np.abs(xf - ref).max())}
    x.scatter_rev_begin()
    x.scatter_rev_end()
    torch.cuda.synchronize()
    xr = x.data_copy()
    ref2 = ref.copy()
    np.add.at(ref2, send, ref[n_local + recv])
    out["rev"] = float(np.abs(xr - ref2).max())
    return out


def _segments_worker(rank, world, port, *rest):
    _reporting(_segments_body)(rank, world, port, *rest)


@pytest.mark.parametrize("halo", ["exchange", "windows"])
def test_native_exchange_with_several_segments(built, halo):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (out,) = _run_ranks(_segments_worker, 1, (halo,))
    assert out["fwd"] == 0.0
    assert out["rev"] < 1e-14


# ---------------------------------------------------------------------------------------------
# Real RCCL between GPUs: runs wherever at least two devices are visible (the one-GPU test box skips it).
# "add one 2-rank test of forward and reverse exchange plus a captured cycle, to run once when a
# multi-GPU box is available".
def _multi_gpu_body(rank, world, port, n, dims, orders, halo="exchange"):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        import pmg_dolfinx_amd as pm

        comm = pm.RcclComm.from_torch(device=torch.device("cuda", rank), halo=halo)
        assert comm.size() == world
        out = _rank_checks(pm, rank, world, n, dims, orders, comm=comm)
        # forward + reverse scatter through the communicator against the partition's own lists
        H = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=3, proc_dims=dims, rank=rank, size=world, warp=warp,
                                comm=comm)
        lv, lay = H.levels[-1], H.layouts[-1]
        v = pm.Vector(lay)
        g = np.arange(H.part.global_ndofs(orders[-1]), dtype=np.float64)
        loc = np.zeros(lv.ndofs)
        loc[: lv.size_local] = g[lv.local_to_global[: lv.size_local]]
        v.data.copy_(torch.from_numpy(loc))
        v.scatter_fwd()
        out["fwd_ok"] = bool(np.array_equal(v.data_copy(), g[lv.local_to_global]))
        v.scatter_rev()
        mult = np.bincount(np.concatenate(dist_all_ghost_globals(dist, lv)), minlength=g.size) + 1.0
        out["rev_err"] = float(np.abs(v.data_copy()[: lv.size_local]
                                      - (g * mult)[lv.local_to_global[: lv.size_local]]).max())
        # eager cycles, then the same cycles replayed as a hipGraph with the exchange captured
        def cycles(graph):
            H.mg.set_graph(graph)
            x = H.new_vector()
            x.set(0.0)
            for _ in range(4):
                H.mg.apply(H.rhs[-1], x)
            torch.cuda.synchronize()
            return x.data_copy()[: lv.size_local].copy(), H.mg.graph_replays()

        xe, r0 = cycles(False)
        xg, r1 = cycles(True)
        H.mg.set_graph(False)
        out["graph_replays"] = r1 - r0
        out["graph_vs_eager"] = float(np.abs(xg - xe).max() / np.abs(xe).max())
        dist.barrier()  # (windows) nobody frees a window a neighbour may still acknowledge into
        return out
    finally:
        dist.destroy_process_group()


def dist_all_ghost_globals(dist, lv):
    """Global indices of every rank's ghosts (how often each owned dof is ghosted elsewhere)."""
    mine = np.asarray(lv.local_to_global[lv.size_local:], dtype=np.int64)
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, mine)
    return gathered


def _multi_gpu_worker(rank, world, port, n, dims, orders, *rest):
    _reporting(_multi_gpu_body)(rank, world, port, n, dims, orders, *rest)


@pytest.mark.parametrize("halo", ["exchange", "windows"])
@pytest.mark.parametrize("dims,n", [((1, 1, 1), (4, 4, 4)), ((1, 1, 2), (4, 4, 8)), ((2, 2, 2), (6, 6, 6))])
def test_native_rccl_between_gpus(dims, n, halo, built):
    """(1, 1, 1) is the rehearsal of this test's own code on a one-GPU box: a torch nccl group and a library
    communicator of one rank, no halo.  halo = "windows": the exchange as stores into the peer GPUs' windows over
    xGMI, RCCL for the reductions only."""
    import torch

    world = dims[0] * dims[1] * dims[2]
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    res = _run_ranks(_multi_gpu_worker, world, (n, dims, (1, 2, 4), halo), timeout=600)
    if world > 1:
        _assert_rank_results(res)
    for out in res:
        assert out["fwd_ok"] and out["rev_err"] == 0.0
        assert out["graph_replays"] >= 3 and out["graph_vs_eager"] < 1e-12


def _window_timeout_body(rank, world, port):
    """A neighbour that never answers: the waits of the window kernels are bounded, every wave reaches its end, and
    the next scatter of the layout reports what happened instead of the GPU hanging."""
    import ctypes as C

    import torch

    os.environ["PMG_WINDOW_TIMEOUT_MS"] = "200"
    import pmg_dolfinx_amd as pm
    from pmg_dolfinx_amd import _lib
    from pmg_dolfinx_amd.vector import HaloWindows

    torch.cuda.set_device(0)
    n, m = 1000, 100
    # a layout with one neighbour whose window and flags nobody serves
    send = torch.arange(m, dtype=torch.int32, device="cuda")
    h = _lib.vp()
    sbuf, rbuf = torch.zeros(m, dtype=torch.float64, device="cuda"), torch.zeros(m, dtype=torch.float64, device="cuda")
    _lib.call("pmg_layout_create", C.byref(h), n, m, m, _lib.ptr(send), _lib.ptr(sbuf), m, _lib.ptr(send),
              _lib.ptr(rbuf), _lib.EXCHANGE_FN(), _lib.ALLREDUCE_FN(), _lib.vp(0))
    cnt = np.array([m], dtype=np.int32)
    doubles, fwd, rev = C.c_int64(), np.zeros(1, np.int64), np.zeros(1, np.int64)
    _lib.call("pmg_layout_window_describe", 1, cnt.ctypes.data_as(_lib.c_ip), cnt.ctypes.data_as(_lib.c_ip),
              C.byref(doubles), fwd.ctypes.data_as(_lib.c_lp), rev.ctypes.data_as(_lib.c_lp))
    mine = [HaloWindows._alloc(8 * doubles.value)[0], HaloWindows._alloc(8 * HaloWindows.FLAG_WORDS)[0]]
    peer = [HaloWindows._alloc(8 * doubles.value)[0], HaloWindows._alloc(8 * HaloWindows.FLAG_WORDS)[0]]
    nbw, nbf = (_lib.vp * 1)(peer[0].value), (_lib.vp * 1)(peer[1].value)
    nd = np.array([doubles.value], np.int64)
    slot = np.zeros(1, np.int32)
    _lib.call("pmg_layout_set_windows", h, 1, cnt.ctypes.data_as(_lib.c_ip), cnt.ctypes.data_as(_lib.c_ip), mine[0],
              mine[1], nbw, nbf, nd.ctypes.data_as(_lib.c_lp), fwd.ctypes.data_as(_lib.c_lp),
              rev.ctypes.data_as(_lib.c_lp), slot.ctypes.data_as(_lib.c_ip))
    x = torch.zeros(n + m, dtype=torch.float64, device="cuda")
    st = _lib.current_stream()
    import time

    t0 = time.time()
    _lib.call("pmg_scatter_fwd_begin", h, _lib.ptr(x), st)
    _lib.call("pmg_scatter_fwd_end", h, _lib.ptr(x), st)  # waits for an arrival that never comes
    torch.cuda.synchronize()
    out = {"seconds": time.time() - t0}
    try:
        _lib.call("pmg_scatter_fwd_begin", h, _lib.ptr(x), st)
        out["error"] = ""
    except RuntimeError as e:
        out["error"] = str(e)
    torch.cuda.synchronize()
    _lib.lib().pmg_layout_destroy(h)
    for p in mine + peer:
        _lib.lib().pmg_window_free(p)
    return out


def _window_timeout_worker(rank, world, port, q):
    _reporting(_window_timeout_body)(rank, world, port, q)


def test_a_halo_window_wait_is_bounded(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    (out,) = _run_ranks(_window_timeout_worker, 1, ())
    assert 0.15 < out["seconds"] < 5.0, out
    assert "did not arrive within the time limit" in out["error"], out


def _stretch(x):
    """A separable stretch: cell centroids stay on a tensor grid (tensor-block patches, chains of patches)."""
    return x + 0.05 * np.sin(2.0 * np.pi * x)


def test_chain_form_between_two_ranks_as_threads(built, monkeypatch):
    """The chain form of the degree-4 interior launches (include/pmg_amd.h "Chain form") inside a distributed
    application: halo begin, the interior as chains, halo end, the boundary shell as one atomic launch -- two ranks
    as threads, apply and two V-cycles against the single-domain oracle."""
    import threading

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pmg_dolfinx_amd as pm
    from oracle import pmg_oracle as po

    n, dims, orders, world, k = (12, 4, 32), (2, 1, 1), (2, 4), 2, 3
    monkeypatch.setenv("PMG_CHAIN", "2")
    W = _ThreadWorld(world)
    res, errors = [None] * world, []
    gm = po.BoxMesh(n, warp=_stretch)
    A = po.Laplacian(4, 2.0, gm.dofmap(4), gm.xgeom, gm.geom_dofmap, gm.boundary_marker(4))
    ug = np.random.default_rng(3).standard_normal(A.ndofs)
    Aug = A.apply(ug)
    mesh, ops, sm, it, mg, b, eigs = po.build_hierarchy(n, orders, cheb_its=k, warp=_stretch)

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                H = pm.PoissonHierarchy(n, orders, kappa=2.0, cheb_its=k, proc_dims=dims, rank=rank, size=world,
                                        warp=_stretch, comm=_ThreadComm(W, rank))
                lv, layout, op = H.levels[-1], H.layouts[-1], H.operators[-1]
                x, y = pm.Vector(layout), pm.Vector(layout)
                xl = np.zeros(lv.ndofs)
                xl[: lv.size_local] = ug[lv.local_to_global[: lv.size_local]]
                x.data.copy_(torch.from_numpy(xl))
                errs = []
                for form in (True, False):
                    if op.chain_available():
                        op.set_chain_form(form)
                    y.set(9.0)
                    op(x, y)
                    ref = Aug[lv.local_to_global[: lv.size_local]]
                    errs.append(float(np.abs(y.data_copy()[: lv.size_local] - ref).max() / np.abs(ref).max()))
                if op.chain_available():
                    op.set_chain_form(True)
                # two V-cycles against the oracle's, with this hierarchy's smoother bounds (rank 0 runs the oracle)
                if rank == 0:
                    for s_, e in zip(sm, H.eig_ranges):
                        s_.eig_range = e
                    xo = np.zeros_like(b)
                    shared["xo"] = []
                    for _ in range(2):
                        xo = mg.apply(b, xo)
                        shared["xo"].append(xo.copy())
                W.wait()
                xv = H.new_vector()
                xv.set(0.0)
                verr = []
                for c in range(2):
                    H.mg.apply(H.rhs[-1], xv)
                    ref = shared["xo"][c]
                    verr.append(float(np.abs(xv.data_copy()[: lv.size_local] - ref[lv.local_to_global[: lv.size_local]]).max()
                                      / np.abs(ref).max()))
                torch.cuda.current_stream().synchronize()
                res[rank] = {"chains": op.chain_available(), "launches": op.launches_per_apply(), "apply_err": errs,
                             "vcycle_err": verr}
        except BaseException:  # noqa: BLE001
            import traceback

            errors.append((rank, traceback.format_exc()))
            W.barrier.abort()

    shared = {}
    try:
        pm.set_merge_threshold(0)  # coloured launches on this small mesh too
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
    finally:
        pm.set_merge_threshold(-1)
    assert not errors, "\n".join(f"rank {r}:\n{tb}" for r, tb in errors)
    assert all(r is not None for r in res)
    assert any(r["chains"] for r in res), "no rank built chains: the test does not exercise the chain form"
    for r in res:
        assert max(r["apply_err"]) < 1e-12, r
        assert max(r["vcycle_err"]) < 1e-10, r
