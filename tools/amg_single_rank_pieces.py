"""GPU time of two stationary AMG cycles on ONE rank (no transport in the way): the fully replicated form and the
form with the fine level on the (here: unpartitioned) matrix-free operator, at n^3 degree-1 cells.
usage: python tools/amg_single_rank_pieces.py n [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm

for n in [int(a) for a in sys.argv[1:]] or [64]:
    part = pm.BoxPartition(n)
    lv = part.level(1)
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
    op.compute_diag_inverse()
    g = np.random.default_rng(3).standard_normal(lv.ndofs)
    g[lv.bc_marker.astype(bool)] = 0.0
    b, x = pm.Vector(layout), pm.Vector(layout)
    b.data.copy_(torch.from_numpy(g))
    amg = pm.AmgSolver(op, cycles=2, global_index=np.arange(lv.ndofs), n_global=lv.ndofs)
    rows = [l["rows"] for l in amg.info()]
    for name, flag in (("fully replicated", 0), ("fine level on the matrix-free operator", 1)):
        pm._lib.call("pmg_amg_set_distributed_fine_level", amg.handle, flag)
        for _ in range(3):
            amg.solve(x, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            amg.solve(x, b)
        e1.record()
        e1.synchronize()
        print(f"{n}^3 cells, rows {rows}: two stationary cycles, {name}: {e0.elapsed_time(e1) / 20:.3f} ms  |x| = {pm.norm(x):.10e}")
    del amg, op, x, b
    torch.cuda.empty_cache()
