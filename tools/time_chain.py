"""Degree-4 operator application: chain form against the patch-colour launches, same operator, alternating rounds.
usage: PMG_CHAIN=1 python tools/time_chain.py [n] [reps] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMG_CHAIN", "1")
import numpy as np, torch
import pmg_dolfinx_amd as pm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P = 4
part = pm.BoxPartition(n); lv = part.level(P); layout = pm.make_layout(lv)
op = pm.MatFreeLaplacian(P, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells, lv.bc_marker, layout)
print("chains available:", op.chain_available(), flush=True)
x, y = pm.Vector(layout), pm.Vector(layout)
x.data.copy_(torch.randn(lv.ndofs, dtype=torch.float64, device="cuda"))
N, U = (P + 1) ** 3, P ** 3
alg = (52 * N + 8 + 17 * U) * part.ncells
res = {}
for form in ([True, False] if op.chain_available() else [False]):
    op.set_chain_form(form) if op.chain_available() else None
    op(x, y); torch.cuda.synchronize()
    res[form] = y.data_copy()
if len(res) == 2:
    print("max |chain - patches| / max|y| =", np.abs(res[True] - res[False]).max() / np.abs(res[False]).max(), flush=True)
for r in range(rounds):
    for form in ([True, False] if op.chain_available() else [False]):
        if op.chain_available():
            op.set_chain_form(form)
        op.time_kernel(x, y, 3)
        ms = op.time_kernel(x, y, reps) * op.launches_per_apply()
        print(f"round {r} {'chain  ' if form else 'patches'} launches {op.launches_per_apply()}  {ms*1e3:.1f} us  "
              f"{alg/ms/1e6:.0f} GB/s ({alg/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
