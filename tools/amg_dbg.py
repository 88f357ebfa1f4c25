import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pmg_dolfinx_amd as pm
for n in (24, 48, 64):
    part = pm.BoxPartition(n); lv = part.level(1); lay = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, lay)
    g = np.random.default_rng(3).standard_normal(lv.ndofs); g[lv.bc_marker.astype(bool)] = 0.0
    b = pm.Vector(lay); b.data.copy_(torch.from_numpy(g))
    for setup in ("gathered", "distributed"):
        amg = pm.AmgSolver(op, max_iter=100, rtol=1e-8, global_index=np.arange(lv.ndofs), n_global=lv.ndofs, setup=setup)
        x = pm.Vector(lay)
        its = amg.solve(x, b)
        print(n, setup, its, [(l["rows"], round(l["lambda_max"], 4)) for l in amg.info()], flush=True)
        del amg
