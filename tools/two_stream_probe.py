"""Can a second stream fill the launch tails of the coloured apply?  Two operators (two meshes of n^3 cells, degree P)
applied back to back on one stream against the same two applications issued on two streams.
usage: python tools/two_stream_probe.py P 64x64x64 32x64x64   (the two meshes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmg_dolfinx_amd as pm

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shape = lambda t: tuple(int(v) for v in t.split("x"))  # noqa: E731
sa = shape(sys.argv[2]) if len(sys.argv) > 2 else (64, 64, 64)
sb = shape(sys.argv[3]) if len(sys.argv) > 3 else sa


def make(shape):
    part = pm.BoxPartition(shape)
    lv = part.level(P)
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
    x, y = pm.Vector(layout), pm.Vector(layout)
    x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
    return op, x, y


a, b = make(sa), make(sb)
print("streams per operator:", a[0].apply_streams(), b[0].apply_streams())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = 40


def run(two):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    if two:
        s2.wait_stream(s1)
    for _ in range(reps):
        with torch.cuda.stream(s1):
            a[0](a[1], a[2])
        with torch.cuda.stream(s2 if two else s1):
            b[0](b[1], b[2])
    if two:
        s1.wait_stream(s2)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for _ in range(2):
    one, two = run(False), run(True)
print(f"P={P} meshes {sa} and {sb}: one stream {one:.1f} us per pair, two streams {two:.1f} us per pair "
      f"({two / one:.3f})")
