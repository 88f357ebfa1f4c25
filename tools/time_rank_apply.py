"""Compute cost of ONE rank's share of a multi-GPU run, measured on a single GPU:
the brick of rank R of a px x py x pz partition (owned cells + ghost layer,
interior / boundary patch launches), halo exchange left out.
usage: python tools/time_rank_apply.py P n_per_rank px py pz rank"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm
P, n = int(sys.argv[1]), int(sys.argv[2])
dims = tuple(int(v) for v in sys.argv[3:6]); rank = int(sys.argv[6])
part = pm.BoxPartition(tuple(n * d for d in dims), dims, rank)
lv = part.level(P)
layout = pm.Layout(lv.size_local, lv.num_ghosts)  # ghosts kept, no neighbours: compute only
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
op.time_kernel(x, y, 3)
nl = op.launches_per_apply()
ms = op.time_kernel(x, y, 20) * nl
print(f"rank {rank} of {dims}: cells owned {part.ncells_owned} + ghost {part.ncells - part.ncells_owned}, "
      f"lcells {len(lv.lcells)} bcells {len(lv.bcells)}, launches {nl}, apply kernels {ms*1e3:.1f} us "
      f"({ms*1e3/part.ncells_owned*64**3:.1f} us per 64^3 owned cells)")
