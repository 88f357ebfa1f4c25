"""Cost of the replicated AMG coarse solve with and without its fine level distributed, at BASELINE config 5's
weak-scaled size: the degree-1 level of (2n)^3 cells over 2x2x2 ranks (n^3 cells each), the ranks being eight host
threads that share the one GPU (the in-process transport of tests/test_gpu_distributed.py).  All ranks solve at the
same time, so the wall time of a solve divided by the rank count estimates one rank's GPU time.
usage: python tools/amg_dist_fine_level.py [n_per_rank]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pmg_dolfinx_amd as pm
from test_gpu_distributed import _ThreadComm, _ThreadWorld

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dims = (2, 2, 2)
world = 8
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
            op.compute_diag_inverse()
            g = np.random.default_rng(3).standard_normal(part.global_ndofs(1))[lv.local_to_global]
            g[lv.bc_marker.astype(bool)] = 0.0
            b, x = pm.Vector(layout), pm.Vector(layout)
            b.data.copy_(torch.from_numpy(g))
            out = {}
            t0 = time.perf_counter()
            amg = pm.AmgSolver(op, cycles=2, global_index=lv.local_to_global, n_global=part.global_ndofs(1))
            torch.cuda.current_stream().synchronize()
            out["setup_s"] = time.perf_counter() - t0
            out["levels"] = [l["rows"] for l in amg.info()]
            for name, flag in (("distributed fine level", 1), ("fully replicated", 0)):
                pm._lib.call("pmg_amg_set_distributed_fine_level", amg.handle, flag)
                for _ in range(2):
                    amg.solve(x, b)
                torch.cuda.current_stream().synchronize()
                W.barrier.wait()
                t0 = time.perf_counter()
                reps = 10
                for _ in range(reps):
                    amg.solve(x, b)
                torch.cuda.current_stream().synchronize()
                W.barrier.wait()
                out[name] = (time.perf_counter() - t0) / reps
                out[name + " |x|"] = float(pm.norm(x))
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
r0 = res[0]
print(f"{n}^3 cells of degree 1 over 2x2x2 ranks ({nr}^3 each), hierarchy rows {r0['levels']}, set-up {r0['setup_s']:.1f} s "
      f"(eight set-ups at once on one host)")
for name in ("fully replicated", "distributed fine level"):
    wall = max(r[name] for r in res)
    print(f"  two stationary cycles, {name:24s}: {wall * 1e3:7.2f} ms wall for 8 ranks on one GPU = "
          f"{wall * 1e3 / world:5.2f} ms of GPU time per rank   (|x| = {r0[name + ' |x|']:.10e})")
