"""Time the stiffness kernel alone (HIP events) for one degree / mesh size.
usage: python tools/time_apply.py P n [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm
P, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
part = pm.BoxPartition(n); lv = part.level(P); layout = pm.make_layout(lv)
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
class V: pass
big = part.ncells * (P + 1) ** 3 * 2 + 4096  # room for the linear gather / write-back ablations
x = V(); x.data = torch.randn(big, dtype=torch.float64, device="cuda")
v = V(); v.data = torch.zeros(big, dtype=torch.float64, device="cuda")
op.time_kernel(x, v, 3)
ms = op.time_kernel(x, v, reps) * op.launches_per_apply()
N, U = (P + 1) ** 3, P ** 3
alg = (52 * N + 8 + 17 * U) * part.ncells
if os.environ.get("PMG_GEOMETRY") == "affine":
    op.set_geometry_mode("affine")
    op.time_kernel(x, v, 3)
    ms = op.time_kernel(x, v, reps) * op.launches_per_apply()
print(f"geom={os.environ.get('PMG_GEOMETRY','stored')} lib={os.environ.get('PMG_AMD_LIB','default')} P={P} n={n} kernel {ms*1e3:.1f} us  algorithmic {alg/ms/1e6:.0f} GB/s  ({alg/ms/1e6/8000:.3f} of 8 TB/s)")
