"""Diagnostic (needs a -DPMG_STAMPS build, tools/build_ablation.sh stamps "-DPMG_STAMPS"):
wall-clock phase durations inside the workgroups of the column stiffness kernel.
usage: PMG_AMD_LIB=tools/abl/libpmg_amd_stamps.so python tools/stamp_phases.py [P n cells_per_patch]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm
P, n, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 64, 32)))
part = pm.BoxPartition(n); lv = part.level(P); layout = pm.make_layout(lv)
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
npatch = part.ncells // K
buf = torch.zeros(npatch * 4, dtype=torch.int64, device="cuda")
L = pm._lib.lib()
L.pmg_debug_set_stamp_buffer.argtypes = [C.c_void_p]
assert L.pmg_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
for _ in range(3):
    op(x, y)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(-1, 4).astype(np.float64) / 100.0  # us
d = np.diff(s, axis=1)
tot = s[:, 3] - s[:, 0]
print("patches", len(s), " per-workgroup us: gather %.2f  cells %.2f  write-back+drain %.2f  total %.2f (median)" % (
    np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.median(tot)))
print("mean: gather %.2f cells %.2f wb %.2f total %.2f ; p90 total %.2f" % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), tot.mean(), np.percentile(tot, 90)))
# per launch (colour): span from first start to last end, and sum of workgroup time / (512 slots)
per = npatch // 8
for c in range(8):
    blk = s[c * per:(c + 1) * per]
    span = blk[:, 3].max() - blk[:, 0].min()
    print(f"colour {c}: span {span:7.1f} us, workgroup-time/512 slots {tot[c*per:(c+1)*per].sum()/512:7.1f} us, first starts within {np.percentile(blk[:,0]-blk[:,0].min(), 50):6.1f} us (median)")
