"""Does the alignment of a neighbour's segment inside the staging buffers matter to RCCL?  One rank, its own partner
through the library's communicator, TWO (or seven) segments to itself; the first segment's length decides the alignment
of the rest.   usage: python tools/time_halo_alignment.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm

torch.cuda.set_device(0)
comm = pm.RcclComm(0, 1, pm.RcclComm.unique_id())
n_local, n_ghost = 2_000_000, 264_192


def timed(counts, reps=200):
    m = sum(counts)
    send = np.arange(m, dtype=np.int32) * 7 % n_local
    lay = pm.Layout(n_local, n_ghost, [0] * len(counts), counts, counts, send, np.arange(m, dtype=np.int32), comm=comm)
    x = pm.Vector(lay)
    x.data.normal_()
    for _ in range(10):
        x.scatter_fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x.scatter_fwd()
    e1.record()
    torch.cuda.synchronize()
    ok = bool(torch.equal(x.data[n_local:n_local + m], x.data[torch.as_tensor(send.astype(np.int64), device="cuda")]))
    return e0.elapsed_time(e1) / reps * 1e3, ok


for name, counts in (("1 segment", [264192]),
                     ("2 segments, second 256-B aligned", [132096, 132096]),
                     ("2 segments, second 8-B aligned", [132095, 132097]),
                     ("7 segments, 256-B aligned", [37728] * 7),
                     ("7 segments, 8-B aligned", [37727, 37729, 37725, 37731, 37723, 37733, 37724])):
    us, ok = timed(counts)
    print(f"{name:36s} {us:8.1f} us per exchange (pack + group + unpack), ghosts correct: {ok}", flush=True)
