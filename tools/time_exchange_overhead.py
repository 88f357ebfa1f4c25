"""Cost of the halo path on ONE GPU: the config-2 V-cycle of rank 0 of a 1x1x2 split (64^3 owned cells plus
the ghost layer; every scatter packs / exchanges / unpacks the real halo volume, ~2 MB at p4) with the rank as
its own partner, through (a) the library's RCCL communicator, (b) the library's halo windows, (c) callbacks into
torch.distributed, against (d) the same brick without any exchange.  The numerics are meaningless (ghosts receive the wrong owned values);
the timings are not: host issue time per cycle, cycle time, i.e. what the exchange costs before any xGMI
link is involved.   usage: python tools/time_exchange_overhead.py [n]   (n^3 owned cells, default 64; 32 = config 3's
per-GPU share on eight GPUs)"""
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


def self_layout(comm_factory):
    def make(lv, group=None, device="cuda", comm=None):
        m = min(sum(lv.send_counts), sum(lv.recv_counts))
        c = comm_factory()
        return pm.Layout(lv.size_local, lv.num_ghosts, [0] if m else [], [m] if m else [], [m] if m else [],
                         lv.send_indices[:m], lv.recv_indices[:m], device=device, comm=c,
                         always_exchange=c is None)
    return make


orig = problem.make_layout
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 64
build = lambda: pm.PoissonHierarchy((NC, NC, 2 * NC), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)

problem.make_layout = lambda lv, group=None, device="cuda", comm=None: pm.Layout(lv.size_local, lv.num_ghosts, device=device)
H = build()
print("ghost dofs per level:", [lv.num_ghosts for lv in H.levels], " halo bytes at p4: %.2f MB" % (H.levels[-1].num_ghosts * 8e-6))
print("no exchange:             %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
del H
native = pm.RcclComm(0, 1, pm.RcclComm.unique_id())
problem.make_layout = self_layout(lambda: native)
H = build()
print("library RCCL communicator: %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
H.mg.set_graph(True)
print("  the same through a hipGraph (exchange captured on the compute stream): %.3f ms per cycle (host issue time %.3f ms), %d replays"
      % (*cycle_ms(H), H.mg.graph_replays()))
H.mg.set_graph(False)
del H
windows = pm.RcclComm(0, 1, pm.RcclComm.unique_id(), halo="windows")
problem.make_layout = self_layout(lambda: windows)
H = build()
print("halo windows, as created (graph mode auto): %.3f ms per cycle (host issue time %.3f ms), %d replays"
      % (*cycle_ms(H), H.mg.graph_replays()))
H.mg.set_graph(False)
print("halo windows (direct stores + flags), eager: %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
H.mg.set_graph(True)
print("  the same through a hipGraph: %.3f ms per cycle (host issue time %.3f ms), %d replays"
      % (*cycle_ms(H), H.mg.graph_replays()))
H.mg.set_graph(False)
del H
problem.make_layout = self_layout(lambda: None)
H = build()
print("torch.distributed callbacks: %.3f ms per cycle (host issue time %.3f ms)" % cycle_ms(H))
del H
problem.make_layout = orig
dist.destroy_process_group()
