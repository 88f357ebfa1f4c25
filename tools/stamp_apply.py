"""Where a wavefront of the apply kernel spends its time: run a library built with -DPMG_STAMPS
(tools/build_variant.sh stamps -DPMG_STAMPS ...) and summarise the per-wavefront clock readings.

usage: PMG_AMD_LIB=tools/abl/lib_stamps_col.so python tools/stamp_apply.py [P] [n] [launch]

Stamps (10 ns ticks of the constant 100 MHz clock): 0 entry, 1 gathered values in LDS, 2 behind the gather's barrier,
3 start of the wavefront's last item, 4 cell loop done, 5 behind the barrier that ends the accumulation, 6 stores
issued, 7 stores acknowledged.  Reported: medians over the wavefronts of ONE colour launch of one application (us),
and the launch's own span.  Read the SHARES, not the total: a stamped build forbids overlaps the product has."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

os.environ.setdefault("PMG_AMD_LIB_ALLOW_MISSING", "1")
import pmg_dolfinx_amd as pm

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
part = pm.BoxPartition(n)
lv = part.level(P)
layout = pm.make_layout(lv)
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
L = pm._lib.lib()
nw = {1: 8, 2: 8, 3: 4, 4: 8, 5: 4, 6: 4, 7: 8, 8: 4}[P]
nwg = 1 << 14
buf = torch.zeros((nwg, nw, 8), dtype=torch.int64, device="cuda")
f = L.pmg_debug_set_stamp_buffer
f.argtypes = [C.c_void_p, C.c_int]
f.restype = C.c_int
for _ in range(3):
    op(x, y)
torch.cuda.synchronize()
assert f(C.c_void_p(buf.data_ptr()), nwg) == 0
op(x, y)  # every colour launch overwrites the records of blockIdx 0 ..: the LAST colour launch of the application remains
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.float64)
used = s[:, 0, 0] > 0
s = s[used]
print(f"lib={os.environ.get('PMG_AMD_LIB','default')} P={P} n={n}: {s.shape[0]} workgroups x {nw} wavefronts in the last "
      f"colour launch")
t0 = s[:, :, 0].min()
tick = 0.01  # us
d = lambda a, b: (s[:, :, b] - s[:, :, a]).ravel() * tick  # noqa: E731
names = [("gather: entry -> values in LDS", 0, 1), ("wait at the gather's barrier", 1, 2),
         ("cell loop: first item(s)", 2, 3), ("cell loop: last item", 3, 4), ("cell loop (all)", 2, 4),
         ("wait at the closing barrier", 4, 5), ("write-back: issue", 5, 6), ("write-back: acknowledgement", 6, 7),
         ("whole wavefront", 0, 7)]
for name, a, b in names:
    v = d(a, b)
    print(f"  {name:38s} median {np.median(v):7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  us")
wg = (s[:, :, 7].max(axis=1) - s[:, :, 0].min(axis=1)) * tick
print(f"  workgroup lifetime                     median {np.median(wg):7.2f}  p10 {np.percentile(wg, 10):7.2f}  "
      f"p90 {np.percentile(wg, 90):7.2f}  us")
print(f"  launch span (first entry -> last acknowledgement) {(s[:, :, 7].max() - t0) * tick:8.2f} us; start times: "
      f"median {np.median((s[:, 0, 0] - t0) * tick):.2f}, p90 {np.percentile((s[:, 0, 0] - t0) * tick, 90):.2f} us")
st = np.sort((s[:, 0, 0] - t0) * tick)
en = np.sort((s[:, :, 7].max(axis=1) - t0) * tick)
q = [1, 25, 49, 51, 75, 99]
print("  workgroup start times, percentiles " + str(q) + ": " + " ".join(f"{np.percentile(st, x):.1f}" for x in q) + " us")
print("  workgroup end times,   percentiles " + str(q) + ": " + " ".join(f"{np.percentile(en, x):.1f}" for x in q) + " us")
# how many workgroups are in their cell loop at a time (of the 512 resident slots)
ts = np.linspace(s[:, :, 0].min(), s[:, :, 7].max(), 200)
incell = [(np.sum((s[:, :, 2] <= t) & (s[:, :, 4] > t))) for t in ts]
alive = [(np.sum((s[:, :, 0] <= t) & (s[:, :, 7] > t))) for t in ts]
print(f"  wavefronts alive: mean {np.mean(alive):.0f}, in their cell loop: mean {np.mean(incell):.0f} "
      f"(of {256 * 2 * nw} resident slots)")
