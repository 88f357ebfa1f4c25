"""Latency of a reduction through the communicator made of windows: `inner_product` of a short vector (the dot
kernel, the fold, the put / get pair of the all-reduce, the host's read of the value) with 1, 2 and 4 ranks that
are PROCESSES SHARING THE ONE GPU, and the sum of a whole level vector (274 625 doubles: 17 chunks) as the replicated
coarse solve does it.  No xGMI link is involved: this prices the kernels and the flag protocol.
usage: python tools/time_window_allreduce.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, q):
    import numpy as np, torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pmg_dolfinx_amd as pm
    from pmg_dolfinx_amd import _lib
    import ctypes as C
    torch.cuda.set_device(0)
    comm = pm.WindowComm.from_torch()
    lay = pm.Layout(1000, 0, comm=comm)
    x = pm.Vector(lay)
    x.set(1.0)
    for _ in range(20):
        pm.inner_product(x, x)
    dist.barrier()
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        v = pm.inner_product(x, x)
    t_dot = (time.perf_counter() - t0) / reps
    assert abs(v - 1000.0 * world) < 1e-9
    # a whole level vector through host_allreduce's device path: use the AMG-style call via the C ABI reduction of slots
    big = torch.ones(274625, dtype=torch.float64, device="cuda")
    lib = _lib.lib()
    lib.pmg_comm_allreduce_sum.restype = C.c_int
    lib.pmg_comm_allreduce_sum.argtypes = [_lib.vp, _lib.vp, C.c_int, _lib.vp]
    st = _lib.current_stream()
    for _ in range(3):
        assert lib.pmg_comm_allreduce_sum(comm.native, _lib.ptr(big), big.numel(), st) == 0
        big.fill_(1.0)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        lib.pmg_comm_allreduce_sum(comm.native, _lib.ptr(big), big.numel(), st)
        big.fill_(1.0)
    torch.cuda.synchronize()
    t_big = (time.perf_counter() - t0) / 20
    dist.barrier()
    if rank == 0:
        print("window memory fine-grained:", _lib.lib().pmg_window_fine_grained(), flush=True)
        q.put((world, t_dot, t_big))
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world in (1, 2, 4):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in ps]
        w, t_dot, t_big = q.get(timeout=300)
        [p.join(timeout=60) for p in ps]
        print(f"{w} rank(s) on one GPU: inner_product (dot + all-reduce + host read) {t_dot*1e6:7.1f} us; "
              f"sum of 274 625 doubles over the ranks {t_big*1e6:7.1f} us", flush=True)
