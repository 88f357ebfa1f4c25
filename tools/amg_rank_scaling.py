"""Krylov iterations of the two multi-rank forms of the AMG coarse solver: the rank-local hierarchy (a block
preconditioner over the ranks) and the replicated one (global matrix gathered on every rank).
The degree-1 Poisson problem on n^3 cells split over 1, 2, 4, 8 ranks (host threads of one process on one GPU,
the in-process transport of tests/test_gpu_distributed.py), solved by CG preconditioned by one AMG cycle to
rtol 1e-8; prints the iteration counts.   usage: python tools/amg_rank_scaling.py [n]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pmg_dolfinx_amd as pm
from test_gpu_distributed import _ThreadComm, _ThreadWorld

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
torch.cuda.set_device(0)
for dims in ((1, 1, 1), (1, 1, 2), (1, 2, 2), (2, 2, 2)):
    world = dims[0] * dims[1] * dims[2]
    W = _ThreadWorld(world)
    its, errs = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                part = pm.BoxPartition(n, dims, rank)
                lv = part.level(1)
                comm = _ThreadComm(W, rank) if world > 1 else None
                layout = pm.Layout(lv.size_local, lv.num_ghosts, lv.neighbors, lv.send_counts, lv.recv_counts,
                                   lv.send_indices, lv.recv_indices, comm=comm)
                op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                                         lv.bc_marker, layout)
                op.compute_diag_inverse()
                amg = pm.AmgSolver(op, max_iter=200, rtol=1e-8)
                rep = pm.AmgSolver(op, max_iter=200, rtol=1e-8, global_index=lv.local_to_global,
                                   n_global=part.global_ndofs(1))
                g = np.random.default_rng(3).standard_normal(part.global_ndofs(1))[lv.local_to_global]
                g[lv.bc_marker.astype(bool)] = 0.0
                b, x = pm.Vector(layout), pm.Vector(layout)
                b.data.copy_(torch.from_numpy(g))
                its[rank] = (amg.solve(x, b), rep.solve(x, b))
                torch.cuda.current_stream().synchronize()
        except BaseException:
            import traceback
            errs.append(traceback.format_exc())
            W.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        print(errs[0])
        sys.exit(1)
    print(f"{n}^3 cells, {dims[0]}x{dims[1]}x{dims[2]} ranks, CG + AMG to rtol 1e-8: rank-local hierarchy {its[0][0]} "
          f"iterations, replicated hierarchy {its[0][1]} iterations")
