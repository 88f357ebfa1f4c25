"""Does the dof NUMBERING limit the operator apply?  ("patch-major vector ordering -- measure".)
The kernels are numbering-agnostic (everything goes through the dofmap), so the experiment is a permutation of the
mesh's dof numbers before the operator is built: lexicographic over the brick (what the mesh generator produces:
a patch's gather / write-back moves runs of P*bz+1 consecutive dofs) against patch-major (the dofs a patch touches
first are one contiguous range, in the order the patch's cells list them).
usage: python tools/time_apply_numbering.py P n [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pmg_dolfinx_amd as pm

P, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
SHAPE = {1: (4, 4, 8), 2: (4, 4, 8), 3: (2, 2, 8), 4: (2, 2, 8), 5: (2, 2, 4), 6: (2, 2, 2), 7: (2, 2, 2), 8: (1, 1, 4)}
part = pm.BoxPartition(n)
lv = part.level(P)
N = (P + 1) ** 3


def patch_major(dofmap, ndofs):
    xg = np.asarray(part.xgeom).reshape(-1, 3)
    cen = xg[np.asarray(part.geom_dofmap).reshape(part.ncells, -1)].mean(axis=1)
    ijk = np.minimum((cen * n).astype(np.int64), n - 1)
    b = np.array(SHAPE[P])
    pid = ijk // b
    loc = ijk % b
    npb = -(-n // b)
    key = ((pid[:, 0] * npb[1] + pid[:, 1]) * npb[2] + pid[:, 2]) * b.prod() + (loc[:, 0] * b[1] + loc[:, 1]) * b[2] + loc[:, 2]
    order = np.argsort(key, kind="stable")
    flat = np.asarray(dofmap).reshape(part.ncells, N)[order].ravel()
    _, first = np.unique(flat, return_index=True)      # first touch of every dof in patch order
    touched = flat[np.sort(first)]                     # old ids in first-touch order
    perm = np.empty(ndofs, dtype=np.int64)
    perm[touched] = np.arange(touched.size)
    assert touched.size == ndofs
    return perm


def time(dofmap, bc, label):
    layout = pm.make_layout(lv)
    op = pm.MatFreeLaplacian(P, 2.0, dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, bc, layout)
    class V: pass
    x = V(); x.data = torch.randn(lv.ndofs, dtype=torch.float64, device="cuda")
    y = V(); y.data = torch.zeros(lv.ndofs, dtype=torch.float64, device="cuda")
    op.time_kernel(x, y, 3)
    ms = op.time_kernel(x, y, reps) * op.launches_per_apply()
    alg = (52 * N + 8 + 17 * P ** 3) * part.ncells
    print(f"P={P} n={n} {label:>14}: kernel {ms*1e3:.1f} us  algorithmic {alg/ms/1e6:.0f} GB/s  ({alg/ms/1e6/8000:.3f} of 8 TB/s)")
    return op, x, y


dm = np.asarray(lv.dofmap).reshape(part.ncells, N)
bc = np.asarray(lv.bc_marker)
op0, x0, y0 = time(dm, bc, "lexicographic")
perm = patch_major(dm, lv.ndofs)
dm2 = perm[dm].astype(dm.dtype)
bc2 = np.empty_like(bc); bc2[perm] = bc
op1, x1, y1 = time(dm2, bc2, "patch-major")
# same operator: y1[perm] == y0 for x1[perm] = x0
pd = torch.from_numpy(perm).cuda()
x1.data[pd] = x0.data
op0(x0, y0)
op1(x1, y1)
torch.cuda.synchronize()
err = float((y1.data[pd] - y0.data).abs().max() / y0.data.abs().max())
print(f"same operator under the renumbering: max relative difference {err:.1e}")
