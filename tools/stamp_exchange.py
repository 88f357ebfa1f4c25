"""Timeline of the whole-exchange window kernel (library built with -DPMG_STAMPS: tools/build_variant.sh wstamps
-DPMG_STAMPS), one rank as its own halo partner, 32^3 owned cells: thread 0's clock readings for one operator
application per level.   usage: PMG_AMD_LIB=tools/abl/lib_wstamps.so python tools/stamp_exchange.py [n]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMG_AMD_LIB_ALLOW_MISSING", "1")
import numpy as np
import torch
import pmg_dolfinx_amd as pm
from pmg_dolfinx_amd import problem

NC = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.cuda.set_device(0)
native = pm.RcclComm(0, 1, pm.RcclComm.unique_id(), halo="windows")


def make(lv, group=None, device="cuda", comm=None):
    m = min(sum(lv.send_counts), sum(lv.recv_counts))
    return pm.Layout(lv.size_local, lv.num_ghosts, [0] if m else [], [m] if m else [], [m] if m else [],
                     lv.send_indices[:m], lv.recv_indices[:m], device=device, comm=native)


problem.make_layout = make
H = pm.PoissonHierarchy((NC, NC, 2 * NC), (1, 2, 4), cheb_its=3, proc_dims=(1, 1, 2), rank=0, size=2)
L = pm._lib.lib()
f = L.pmg_debug_read_window_stamps
f.argtypes = [C.c_void_p]
f.restype = C.c_int
names = ["entry -> sequence number read, slot free", "gather + stores issued", "stores acknowledged (all waves)",
         "release fence", "flags stored", "arrival seen", "window read + ghosts written", "counted out, end"]
for lv, op, lay in zip(H.levels, H.operators, H.layouts):
    x, y = pm.Vector(lay), pm.Vector(lay)
    x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
    rows = []
    for rep in range(6):
        op(x, y)
        torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        assert f(st) == 0
        s = np.array(list(st)[:9], dtype=np.float64)
        rows.append(np.diff(s) * 0.01)
    d = np.median(np.array(rows[2:]), axis=0)
    print(f"degree {lv.P}: {sum(lv.send_counts)} entries out, {sum(lv.recv_counts)} in; whole kernel "
          f"{d.sum():.2f} us of clock readings (thread 0 of block 0)")
    for nme, v in zip(names, d):
        print(f"    {nme:45s} {v:6.2f} us")
