"""Apply timing on a 64^3 p=4 box whose numbering is not the structured one:
'cells' = cells in random order, 'dofs' = dofs renumbered at random (worst case for the
gather / write-back), 'jitter' = vertices off the tensor grid (Morton patch builder).
usage: python tools/time_apply_irregular.py [P n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm

P, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 64)
part = pm.BoxPartition(n)
lv = part.level(P)
rng = np.random.default_rng(0)
N, U = (P + 1) ** 3, P ** 3
alg = (52 * N + 8 + 17 * U) * part.ncells


def run(name, dofmap, xgeom, gdm, bc):
    layout = pm.Layout(lv.ndofs)
    cells = np.arange(part.ncells, dtype=np.int32)
    op = pm.MatFreeLaplacian(P, 2.0, dofmap, xgeom, gdm, cells, cells[:0], bc, layout)
    x, y = pm.Vector(layout), pm.Vector(layout)
    x.data.normal_()
    op.time_kernel(x, y, 3)
    ms = op.time_kernel(x, y, 20) * op.launches_per_apply()
    print(f"{name:28s} launches {op.launches_per_apply():3d}  {ms*1e3:8.1f} us  {alg/ms/1e6:6.0f} GB/s algorithmic ({alg/ms/1e6/8000:.3f})", flush=True)


run("structured", lv.dofmap, part.xgeom, part.geom_dofmap, lv.bc_marker)
cperm = rng.permutation(part.ncells)
run("cells permuted", lv.dofmap[cperm], part.xgeom, part.geom_dofmap[cperm], lv.bc_marker)
xj = part.xgeom + (0.1 / n) * rng.uniform(-1, 1, part.xgeom.shape) * ((part.xgeom > 1e-9) & (part.xgeom < 1 - 1e-9))
run("cells permuted + jitter", lv.dofmap[cperm], xj, part.geom_dofmap[cperm], lv.bc_marker)
# dofs numbered patch-block-wise (a locality-preserving but non-lexicographic numbering)
blk = part.dof_coordinates(P)
key = (np.floor(blk[:, 0] * n / 2) * 1e6 + np.floor(blk[:, 1] * n / 2) * 1e3 + np.floor(blk[:, 2] * n / 8))
order = np.argsort(key, kind="stable")
dperm = np.empty(lv.ndofs, dtype=np.int64); dperm[order] = np.arange(lv.ndofs)
bc = np.zeros(lv.ndofs, np.int8); bc[dperm] = lv.bc_marker
run("dofs block-numbered", dperm[lv.dofmap].astype(np.int32), part.xgeom, part.geom_dofmap, bc)
dperm = rng.permutation(lv.ndofs)
bc = np.zeros(lv.ndofs, np.int8); bc[dperm] = lv.bc_marker
run("dofs random", dperm[lv.dofmap].astype(np.int32), part.xgeom, part.geom_dofmap, bc)
