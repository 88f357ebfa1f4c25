"""Operator application inside a captured graph: 20 applications of one operator captured once, replayed and timed.
usage: PMG_APPLY_STREAMS=0|1 python tools/graph_apply_probe.py [P] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmg_dolfinx_amd as pm

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
part = pm.BoxPartition(n)
lv = part.level(P)
layout = pm.make_layout(lv)
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
s = torch.cuda.Stream()
reps = 20
with torch.cuda.stream(s):
    for _ in range(3):
        op(x, y)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            op(x, y)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        g.replay()
    e1.record(s)
    torch.cuda.synchronize()
    tg = e0.elapsed_time(e1) / 5 / reps * 1e3
    e0.record(s)
    for _ in range(5 * reps):
        op(x, y)
    e1.record(s)
    torch.cuda.synchronize()
    te = e0.elapsed_time(e1) / 5 / reps * 1e3
print(f"P={P} n={n} streams={op.apply_streams()}: application {te:.1f} us eager, {tg:.1f} us in a replayed graph")
