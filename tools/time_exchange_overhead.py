"""Host-side cost of the RCCL exchange path: the config-2 V-cycle on one GPU with every
halo scatter going through torch.distributed (nccl, world size 1, zero-length messages),
against the same cycle without callbacks.  Shows whether the Python callbacks can make
the multi-GPU cycle host-bound.   usage: python tools/time_exchange_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import pmg_dolfinx_amd as pm
from pmg_dolfinx_amd import problem


def cycle_ms(H, reps=20):
    x = H.new_vector()
    x.set(0.0)
    for _ in range(3):
        H.mg.apply(H.rhs[-1], x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        H.mg.apply(H.rhs[-1], x)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, t_issue / reps * 1e3


H = pm.PoissonHierarchy(64, (1, 2, 4), cheb_its=3)
print("no callbacks:   %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
del H
orig = problem.make_layout
problem.make_layout = lambda lv, group=None, device="cuda": pm.Layout(
    lv.size_local, lv.num_ghosts, lv.neighbors, lv.send_counts, lv.recv_counts, lv.send_indices, lv.recv_indices,
    group=group, device=device, always_exchange=True)
H = pm.PoissonHierarchy(64, (1, 2, 4), cheb_its=3)
print("rccl callbacks: %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
dist.destroy_process_group()
