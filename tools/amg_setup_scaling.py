"""Set-up time and iteration counts of the two multi-rank AMG set-ups at BASELINE config 5's weak-scaled size: the
degree-1 level of (2n)^3 cells over 2x2x2 ranks (n^3 cells each), the ranks being eight host threads sharing the one
GPU AND the box's 16-CPU quota (each rank's host-side set-up then has ~2 cores, an eighth of what a rank of a real
8-GPU run has).  "gathered": pmg_amg_create_replicated (global level 0 on every rank); "distributed":
pmg_amg_create_distributed (first coarsening per rank, level 1 gathered).  PMG_AMG_TIMING=1 prints the stages.
usage: python tools/amg_setup_scaling.py [n_per_rank] [one|all]   (one: only rank 0 sets up at a time)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pmg_dolfinx_amd as pm
from test_gpu_distributed import _ThreadComm, _ThreadWorld

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dims, world = (2, 2, 2), 8
n = 2 * nr
torch.cuda.set_device(0)
W = _ThreadWorld(world)
res, errs = [None] * world, []


def run(rank):
    try:
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            part = pm.BoxPartition(n, dims, rank)
            lv = part.level(1)
            layout = pm.Layout(lv.size_local, lv.num_ghosts, lv.neighbors, lv.send_counts, lv.recv_counts,
                               lv.send_indices, lv.recv_indices, comm=_ThreadComm(W, rank))
            op = pm.MatFreeLaplacian(1, 2.0, lv.dofmap, part.xgeom, part.geom_dofmap, lv.lcells, lv.bcells,
                                     lv.bc_marker, layout)
            g = np.random.default_rng(3).standard_normal(part.global_ndofs(1))[lv.local_to_global]
            g[lv.bc_marker.astype(bool)] = 0.0
            b, x = pm.Vector(layout), pm.Vector(layout)
            b.data.copy_(torch.from_numpy(g))
            out = {}
            for setup in ("distributed", "gathered"):
                W.barrier.wait()
                t0 = time.perf_counter()
                amg = pm.AmgSolver(op, max_iter=100, rtol=1e-8, global_index=lv.local_to_global,
                                   n_global=part.global_ndofs(1), setup=setup)
                torch.cuda.current_stream().synchronize()
                out[setup + "_setup_s"] = time.perf_counter() - t0
                out[setup + "_levels"] = [l["rows"] for l in amg.info()]
                out[setup + "_its"] = amg.solve(x, b)
                del amg
            res[rank] = out
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
for setup in ("gathered", "distributed"):
    print(f"{n}^3 cells of degree 1 over 2x2x2 ranks ({nr}^3 each), set-up '{setup}': "
          f"{max(r[setup + '_setup_s'] for r in res):.2f} s (eight set-ups at once on one host, 16 CPUs in all), rows per "
          f"level on rank 0 {res[0][setup + '_levels']}, CG + one cycle to rtol 1e-8: {res[0][setup + '_its']} iterations")
