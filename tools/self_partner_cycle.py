"""The config-2 V-cycle of one rank of a 1x1x2 split (64^3 owned cells + ghost layer) with the rank as its own halo
partner through the library's communicator, for a kernel trace of the multi-rank code path on one GPU
(see tools/time_exchange_overhead.py for what that rehearsal is).
usage: python tools/self_partner_cycle.py [cycles] [graph|eager] [n] [rccl|windows|none]   (n^3 owned cells, default 64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmg_dolfinx_amd as pm
from pmg_dolfinx_amd import problem

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 11
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
torch.cuda.set_device(0)
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 64
halo = sys.argv[4] if len(sys.argv) > 4 else "rccl"
native = None if halo == "none" else pm.RcclComm(0, 1, pm.RcclComm.unique_id(), **({"halo": "windows"} if halo == "windows" else {}))


def make(lv, group=None, device="cuda", comm=None):
    m = min(sum(lv.send_counts), sum(lv.recv_counts))
    if native is None:
        return pm.Layout(lv.size_local, lv.num_ghosts, device=device)
    return pm.Layout(lv.size_local, lv.num_ghosts, [0] if m else [], [m] if m else [], [m] if m else [],
                     lv.send_indices[:m], lv.recv_indices[:m], device=device, comm=native)


problem.make_layout = make
H = pm.PoissonHierarchy((NC, NC, 2 * NC), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)
H.mg.set_graph(graph)
x = H.new_vector()
x.set(0.0)
for _ in range(3):
    H.mg.apply(H.rhs[-1], x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(cycles):
    H.mg.apply(H.rhs[-1], x)
e1.record()
torch.cuda.synchronize()
print(f"{cycles} cycles, {e0.elapsed_time(e1) / cycles:.3f} ms per cycle; stiffness launches per cycle {H.mg.apply_counts()}")
