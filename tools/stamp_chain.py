"""Where a wavefront of the chain kernel spends its time: run a library built with -DPMG_STAMPS
(tools/build_variant.sh stamps -DPMG_STAMPS) and summarise the clock readings of every chain's MIDDLE patch.

usage: PMG_AMD_LIB=tools/abl/lib_stamps.so PMG_CHAIN=1 python tools/stamp_chain.py [n | nx,ny,nz]
(PMG_CHAIN=2 for meshes with fewer chains per colour than compute units: how does a unit stream when the others idle?)

Stamps (10 ns ticks): 0 kernel entry, 1 top of the middle patch, 2 its cell loop done, 3 next patch's values requested,
4 behind barrier #1, 5 next patch's values in LDS, 6 stores issued, 7 end of the chain (stores acknowledged)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("PMG_AMD_LIB_ALLOW_MISSING", "1")
os.environ.setdefault("PMG_CHAIN", "1")
import pmg_dolfinx_amd as pm

n = sys.argv[1] if len(sys.argv) > 1 else "64"
n = tuple(int(v) for v in n.split(",")) if "," in n else int(n)
P = 4
part = pm.BoxPartition(n)
lv = part.level(P)
layout = pm.make_layout(lv)
if os.environ.get("PMG_CHAIN") == "2":
    pm.set_merge_threshold(0)  # coloured launches on small meshes too
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
assert op.chain_available()
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
L = pm._lib.lib()
nw, nwg = 16, 1 << 12
buf = torch.zeros((nwg, nw, 8), dtype=torch.int64, device="cuda")
f = L.pmg_debug_set_stamp_buffer
f.argtypes = [C.c_void_p, C.c_int]
f.restype = C.c_int
for _ in range(3):
    op(x, y)
torch.cuda.synchronize()
assert f(C.c_void_p(buf.data_ptr()), nwg) == 0
op(x, y)  # the LAST chain colour's records remain
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.float64)
s = s[s[:, 0, 0] > 0]
tick = 0.01
print(f"n={n}: {s.shape[0]} chains x {nw} wavefronts in the last colour launch")
d = lambda a, b: (s[:, :, b] - s[:, :, a]).ravel() * tick  # noqa: E731
for name, a, b in [("entry -> top of the middle patch", 0, 1), ("cell loop", 1, 2), ("lists + gather issue", 2, 3),
                   ("wait at barrier #1", 3, 4), ("next patch into LDS (waits for its values)", 4, 5),
                   ("write-back issue", 5, 6), ("top of the patch -> stores issued", 1, 6),
                   ("middle patch -> end of the chain", 6, 7), ("whole chain", 0, 7)]:
    v = d(a, b)
    print(f"  {name:46s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
cl = (s[:, :, 2] - s[:, :, 1]) * tick
print(f"  slowest cell loop of a workgroup: median {np.median(cl.max(axis=1)):.2f} us; fastest: {np.median(cl.min(axis=1)):.2f} us")
t0 = s[:, :, 0].min()
print(f"  launch span {(s[:, :, 7].max() - t0) * tick:.1f} us; chain ends p1/p50/p99: "
      + " ".join(f"{np.percentile((s[:, :, 7].max(axis=1) - t0) * tick, q):.1f}" for q in (1, 50, 99)))
pw = np.median(cl, axis=0)
print("  cell loop by wavefront number (median over the chains): " + " ".join(f"{v:.1f}" for v in pw))
b1 = np.median((s[:, :, 4] - s[:, :, 3]) * tick, axis=0)
print("  wait at barrier #1 by wavefront number:                 " + " ".join(f"{v:.1f}" for v in b1))
