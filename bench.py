#!/usr/bin/env python3
"""bench.py -- DoFs/sec of one p-multigrid V-cycle (Poisson, hex, p = 4) on MI355X.

A "step" is one V-cycle (src/pmg.hpp:56-155 semantics) of the BASELINE.json
config-2 workload: 64^3 hexes per GPU, levels p = 4 -> 2 -> 1, 4th-kind
Chebyshev(3)/Jacobi smoothing on every level, lambda_max from 20 Jacobi-CG
iterations (examples/pmg/main.cpp:306-330), kappa = 2, homogeneous Dirichlet,
RHS = GLL-collocated -kappa lap(sin 2 pi x sin 3 pi y sin 4 pi z).  All inputs are
synthetic and resident in HBM before the timed region.  Multi-GPU: one process
per GPU (torch.distributed / RCCL), brick partition with one ghost-cell layer,
weak scaling (64^3 cells per GPU), forward halo overlapped with interior cells.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
        (N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def algorithmic_bytes_per_cell(P: int) -> int:
    """SURVEY.md 8(d), model storedG: 48N [G] + 4N [dofmap] + 8 [kappa] + 8U [x] + 8U [y] + U [bc]."""
    N, U = (P + 1) ** 3, P**3
    return 48 * N + 4 * N + 8 + 17 * U


def cycle_algorithmic_bytes(orders, ncells, ndofs, k):
    """Algorithmic bytes of one lean V-cycle: per level l > 0 2(k+1)+1 applies and 2(5+8k)+6 vector
    passes, coarsest level k+1 applies and 5+8k passes, transfers = both dofmaps + fine and coarse vectors
    (+ multiplicity in the restriction).  32.8 GB for config 2."""
    total = 0.0
    for i, (P, nd) in enumerate(zip(orders, ndofs)):
        applies = (k + 1) if i == 0 else 2 * (k + 1) + 1
        passes = (5 + 8 * k) if i == 0 else 2 * (5 + 8 * k) + 6
        total += applies * algorithmic_bytes_per_cell(P) * ncells + passes * 8.0 * nd
        if i > 0:
            nf, nc = (P + 1) ** 3, (orders[i - 1] + 1) ** 3
            maps = 4.0 * (nf + nc) * ncells
            total += 2 * maps + 8.0 * (2 * nd + ndofs[i - 1]) + 8.0 * (2 * nd + ndofs[i - 1])
    return total


def oracle_cycle_bytes(orders, ncells, ndofs, k):
    """Algorithmic bytes of the cycle the C oracle (oracle/pmg_oracle.c, orc_vcycle / orc_cheb_solve) REALLY runs --
    what `cpu_baseline.algorithmic_GBs` is computed from.  Counted in its source: finest level 2k+1 operator
    applications (pre-smooth from the current iterate with the residual kept: k+1; post-smooth without its last
    A z: k) and 16k+14 vector passes (two smooths, the correction, the copies in and out); levels between 2k
    applications (their pre-smooth starts from zero: no initial A x) and 16k+8 passes; coarsest level k-1
    applications and 8k+1 passes; transfers as in cycle_algorithmic_bytes.  k = 3: 7 / 6 / 2 applications
    (the lean cycle of SURVEY.md 8d counts 2(k+1)+1 and k+1: 9 / 9 / 4)."""
    total, L = 0.0, len(orders)
    for i, (P, nd) in enumerate(zip(orders, ndofs)):
        if i == L - 1 and L > 1:
            applies, passes = 2 * k + 1, 16 * k + 14
        elif i > 0:
            applies, passes = 2 * k, 16 * k + 8
        else:
            applies, passes = (k - 1, 8 * k + 1) if L > 1 else (k, 8 * k + 5)
        total += applies * algorithmic_bytes_per_cell(P) * ncells + passes * 8.0 * nd
        if i > 0:
            nf, nc = (P + 1) ** 3, (orders[i - 1] + 1) ** 3
            maps = 4.0 * (nf + nc) * ncells
            total += 2 * maps + 8.0 * (2 * nd + ndofs[i - 1]) + 8.0 * (2 * nd + ndofs[i - 1])
    return total


GATE_CYCLES = 3          # V-cycles from x0 = 0 every multi-rank route has to reproduce before it is timed
GATE_TOLERANCE = 1e-10   # relative, iterate after GATE_CYCLES cycles (<= 1e-10 after a V-cycle)
GATE_RNORM_TOLERANCE = 1e-8


def gate_verdict(x_ranks, x_oracle, rn_ranks, rn_oracle, tol=GATE_TOLERANCE, rtol_rn=GATE_RNORM_TOLERANCE):
    """The N-rank cycle against the single-domain C oracle (examples/pmg/main.cpp:362-367 prints the residual per
    cycle; here it is compared): iterate after GATE_CYCLES cycles and the residual norm after every cycle.  Pure
    function (tests/test_bench_launch.py): returns (passed, record)."""
    import numpy as np

    x_ranks, x_oracle = np.asarray(x_ranks), np.asarray(x_oracle)
    rec = {"cycles": len(rn_oracle), "tolerance": tol, "residual_tolerance": rtol_rn}
    if x_ranks.shape != x_oracle.shape or not np.all(np.isfinite(x_ranks)):
        rec.update(passed=False, reason="iterate missing, mis-shaped or not finite")
        return False, rec
    scale = max(float(np.abs(x_oracle).max()), 1e-300)
    err = float(np.abs(x_ranks - x_oracle).max() / scale)
    rn_err = max((abs(a - b) / max(abs(b), 1e-300) for a, b in zip(rn_ranks, rn_oracle)), default=float("inf"))
    ok = bool(err <= tol and rn_err <= rtol_rn and len(rn_ranks) == len(rn_oracle))
    rec.update(passed=ok, iterate_rel_err=err, residual_norm_rel_err=float(rn_err),
               residual_norms=[float(v) for v in rn_ranks], oracle_residual_norms=[float(v) for v in rn_oracle])
    return ok, rec


def strong_routes(world, share_gpu, comm_kind):
    """Transports of the strong-scaling block, in the order they run: (name, communicator, halo, captured, skip
    reason).  RCCL is the default route and the one `strong_scaling.value` comes from; the window routes are measured
    next to it, each behind the same gate.  Ranks that share a GPU (rehearsal) cannot use RCCL at all."""
    no_rccl = "RCCL refuses two ranks on one device (ranks share a GPU: rehearsal)" if share_gpu else (
        "--comm windows: no RCCL in this run" if comm_kind == "windows" else None)
    routes = [("rccl_eager", "rccl", "exchange", False, no_rccl), ("rccl_graph", "rccl", "exchange", True, no_rccl)]
    if no_rccl:
        routes += [("windows_eager", "windows", "windows", False, None), ("windows_graph", "windows", "windows", True, None)]
    else:
        routes += [("windows_graph", "rccl", "windows", True, None)]
    return routes


def finish_line(out, failures):
    """A failed parity check or gate is not an "extra": the line keeps its diagnostics, loses its `value`, and the
    process exits 1 (returns the exit code)."""
    if failures:
        out["parity_failed"] = True
        out.setdefault("parity", {})["failures"] = list(failures)
        out["value"] = None  # a fast wrong answer is not a measurement
        return 1
    return 0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class extra:
    """An extra measurement must never cost the headline line: a failure inside the block is
    logged and recorded under its key instead of propagating.  Only on one rank: with several
    ranks a swallowed failure would leave the others waiting in a collective, so it propagates."""

    def __init__(self, out, key):
        self.out, self.key = out, key

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception) and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            import traceback

            log(f"extra measurement '{self.key}' failed:")
            traceback.print_exception(et, ev, tb, file=sys.stderr)
            self.out[self.key] = {"error": f"{et.__name__}: {ev}"}
            return True
        return False


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=64, help="cells per axis per GPU (weak) / in total (strong)")
    ap.add_argument("--orders", type=str, default="1,2,4")
    ap.add_argument("--cheb", type=int, default=3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-affine", action="store_true", help="skip the extra affine-geometry measurement")
    ap.add_argument("--no-sweep", action="store_true", help="skip the operator-apply sweep over degrees (N = 1 only)")
    ap.add_argument("--extras", action="store_true",
                    help="N > 1: also run the extra measurements (PCG, affine mode); by default only N = 1 does")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1: halo + reductions on the library's own RCCL communicator (default) or through "
                         "torch.distributed callbacks")
    ap.add_argument("--halo", choices=["exchange", "windows"], default="exchange",
                    help="N > 1 with the library's communicator: the halo as grouped ncclSend/ncclRecv (default) or as "
                         "direct stores into the neighbours' IPC-mapped windows (pmg_layout_set_windows; RCCL then "
                         "serves the reductions only).  The windows are validated between processes sharing one GPU "
                         "(tests/test_gpu_distributed.py), not yet between GPUs -- hence opt-in; either way the first "
                         "thing a multi-rank run does is check a forward scatter against the partition's own index map")
    ap.add_argument("--comm", choices=["rccl", "windows"], default="rccl",
                    help="N > 1: the library's communicator on RCCL (default) or made of windows (pmg_comm_create_windows: "
                         "reductions AND halo as direct stores into IPC-mapped device memory, no transport library in "
                         "the data path; validated between processes sharing one GPU, not yet between GPUs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL on a box with fewer GPUs than ranks: all --gpus N ranks run on GPU 0 (needs --comm "
                         "windows; the launcher's group is gloo).  Exercises the whole multi-rank path with real "
                         "inter-process exchanges; the line it prints says n_gpus = 1 and carries a `rehearsal` note -- "
                         "it is not a multi-GPU measurement")
    ap.add_argument("--deadline", type=float, default=-1.0,
                    help="seconds after which a run that has not finished prints a line with value null and what it was "
                         "doing, and exits with code 5 (default: 1500 with several ranks -- a rank that died leaves "
                         "the others waiting in a collective --, off on one rank; 0 = off)")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the strong-scaled config-3 block AND with it the numeric gate against the C oracle "
                         "(the line then says parity 'unverified')")
    ap.add_argument("--corrupt-halo", action="store_true",
                    help="TEST of the gate: the last rank's finest-level halo plan of the strong-scaling block gets the "
                         "middle half of its ghost slots reversed; every route must then fail the gate (value null, exit "
                         "code 1)")
    ap.add_argument("--graph-exchange", action="store_true",
                    help="N > 1 with the library's communicator: additionally time the cycles replayed as a hipGraph "
                         "with the halo exchange captured on the compute stream (one hipGraphLaunch instead of ~115 us "
                         "of host time per exchange; validated on one GPU with a rank as its own partner, "
                         "tests/test_gpu_distributed.py, not yet between GPUs -- hence opt-in)")
    ap.add_argument("--rehearse-comm", action="store_true",
                    help="N = 1: run the multi-rank code path (process group, communicator bootstrap, strong-scaling "
                         "block) with a group of one rank -- a rehearsal on a single-GPU box, not a measurement")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the BASELINE config-5 measurement (p = 6 -> 3 -> 1 with the AMG coarse solve)")
    ap.add_argument("--mesh-sweep", action="store_true",
                    help="N = 1: PCG iteration counts at 32^3 / 64^3 / 96^3 with a random right-hand side")
    return ap.parse_args(argv)


class LaunchError(SystemExit):
    """The run cannot measure what was asked for (rank count, devices): message on stderr, exit code 2,
    nothing on stdout -- never a line for a different number of GPUs than `--gpus`."""

    def __init__(self, msg):
        log("bench.py: " + msg)
        super().__init__(2)


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """Devices this process could use.  `torch.cuda.device_count()` does not initialise the GPU on this image,
    so a parent that only counts may still start child ranks."""
    import torch

    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher (the reference: `srun -n 8`, examples/pmg/submit.sh:29):
    start the N ranks as CHILD processes of a process that has not touched the GPU -- what
    `python -m torch.distributed.run --nproc-per-node N` would do (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in the environment, one rank per GPU), without handing the script's own options to that launcher's
    parser (which reads `--n 32` as an abbreviation of its own options) -- hand rank 0's stdout (the JSON line)
    through and return the first non-zero exit code; a rank that fails takes the others down with it."""
    import subprocess

    have = visible_gpus()
    if args.share_gpu and have >= 1:
        have = args.gpus  # all ranks on GPU 0: a rehearsal, labelled as such in the line
    if have < args.gpus:
        raise LaunchError(f"--gpus {args.gpus} asked for, {have} GPU(s) visible: refusing to run "
                          f"(a {have or 1}-rank measurement must not be labelled {args.gpus} GPUs)")
    port = free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    log(f"bench.py: no launcher in the environment, starting {args.gpus} ranks (127.0.0.1:{port}): " + " ".join(cmd))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]  # EVERY child, every pass (any() would stop at the first one still running)
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            break
        if all(c is not None for c in codes):
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc or next((p.returncode for p in procs if p.returncode), 0)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise LaunchError(f"--gpus {args.gpus}: need at least one")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise LaunchError(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    # The contract is ONE JSON line on stdout.  Libraries underneath write to the process's stdout on their own
    # (RCCL prints a version banner when a communicator is created), so file descriptor 1 is pointed at stderr
    # for the whole run and the line goes out through a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # A run on several ranks that stops making progress (a rank died, a collective nobody else entered) still owes its
    # caller ONE line: after --deadline seconds rank 0 prints what it has -- without a `value` -- and every rank exits.
    watch = {"phase": "start-up", "out": None}
    world_env, rank_env = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    deadline = args.deadline if args.deadline >= 0 else (1500.0 if world_env > 1 else 0.0)

    def _expired():
        msg = f"no result after {deadline:.0f} s; last phase: {watch['phase']}"
        log(f"[rank {rank_env}] DEADLINE: {msg}")
        if rank_env == 0:
            line = dict(watch["out"] or {"metric": "DoFs/sec per p-MG V-cycle (Poisson, hex, p=4)", "unit": "DoF/s",
                                         "n_gpus": world_env, "steps": args.steps, "warmup": args.warmup,
                                         "higher_is_better": True, "dtype": "f64", "data": "synthetic"})
            if line.get("value") is not None:
                line["measured_before_the_run_stopped"] = {"value": line["value"], "ms_per_step": line.get("ms_per_step"),
                                                           "note": "not gated against the oracle: not a result"}
            line.update(value=None, ms_per_step=None, error=msg, incomplete=True)
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        else:
            time.sleep(3.0)
        os._exit(5)

    if deadline > 0:
        import threading

        timer = threading.Timer(deadline, _expired)
        timer.daemon = True
        timer.start()

    import numpy as np
    import torch

    import __graft_entry__ as ge

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus
    if args.share_gpu:
        if args.comm != "windows":
            raise LaunchError("--share-gpu needs --comm windows (RCCL refuses two ranks on one device)")
        local_rank = 0
    if visible_gpus() <= local_rank:
        raise LaunchError(f"rank {rank}: local rank {local_rank} has no GPU ({visible_gpus()} visible); bench.py needs "
                          f"one GPU per rank (there is no CPU fallback of the product path)")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    multi = world > 1 or args.rehearse_comm
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only the one-rank rehearsal gets here without a launcher
            os.environ["MASTER_PORT"] = str(free_port())
        if args.share_gpu:  # control plane only (barriers, max over ranks): the data path is the library's windows
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ctl = "cpu" if args.share_gpu else "cuda"  # where the launcher's group reduces the bench's own scalars
    if rank == 0:
        ge.build()
    if multi:
        dist.barrier()
    import pmg_dolfinx_amd as pm

    orders = tuple(int(p) for p in args.orders.split(","))
    dims = pm.default_proc_dims(world)
    n_global = tuple(args.n * d for d in dims) if args.scaling == "weak" else (args.n,) * 3
    # one process per GPU: the library's RCCL communicator (grouped ncclSend/ncclRecv per halo, ncclAllReduce
    # on device scalars), bootstrapped over the torch.distributed group the launcher gave us
    comm, comm_note = None, None
    if multi and args.exchange == "native":
        try:
            if args.comm == "windows":
                comm = pm.WindowComm.from_torch()
            else:
                comm = pm.RcclComm.from_torch(device=torch.device("cuda", local_rank), halo=args.halo)
        except Exception as e:  # e.g. no librccl to bind: every rank fails alike; the callback route still is RCCL
            comm_note = f"native communicator unavailable ({type(e).__name__}: {e}); torch.distributed callbacks"
            log(f"[rank {rank}] {comm_note}")
        ok = torch.tensor([1 if comm is not None else 0], device=ctl)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None  # all ranks on the same route

    rccl_ranks = (comm.size() if comm is not None else dist.get_world_size()) if multi else 1
    if rccl_ranks != args.gpus:
        raise LaunchError(f"[rank {rank}] the communicator has {rccl_ranks} rank(s), --gpus is {args.gpus}")

    t0 = time.time()
    watch["phase"] = "set-up of the weak-scaled hierarchy"
    H = pm.PoissonHierarchy(n_global, orders, kappa=2.0, cheb_its=args.cheb, proc_dims=dims, rank=rank, size=world,
                            comm=comm)
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.time() - t0:.1f}s dims={dims} n_global={n_global} "
        f"local dofs={[lv.size_local for lv in H.levels]} ghosts={[lv.num_ghosts for lv in H.levels]} "
        f"lmax={[round(e[1], 6) for e in H.eig_ranges]}")
    if multi:
        # the halo moves what the partition says it moves, on every level, before anything is timed: owned entries
        # carry their global index, the ghosts must come back with theirs
        for lv, lay in zip(H.levels, H.layouts):
            v = pm.Vector(lay)
            loc = np.full(lv.ndofs, -1.0)
            loc[: lv.size_local] = lv.local_to_global[: lv.size_local]
            v.data.copy_(torch.from_numpy(loc))
            v.scatter_fwd()
            good = torch.tensor([1 if np.array_equal(v.data_copy(), np.asarray(lv.local_to_global, dtype=np.float64))
                                 else 0], device=ctl)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if int(good.item()) == 0:
                raise LaunchError(f"[rank {rank}] the halo exchange ({args.halo}) does not reproduce the index map")
    P = orders[-1]
    fine_dofs_global = H.part.global_ndofs(P)
    b = H.rhs[-1]
    x = H.new_vector()
    x.set(0.0)

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then EXACTLY K timed V-cycles (stationary iteration, examples/pmg/main.cpp:362-367) ----
    # `value` is the EAGER cycle (stream-ordered launches) on every route; the replayed cycle is reported next to it
    H.mg.set_graph(False)
    watch["phase"] = "warm-up and timed cycles of the headline"
    for _ in range(args.warmup):
        H.mg.apply(b, x)
    sync_all()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        H.mg.apply(b, x)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = fine_dofs_global * args.steps / elapsed
    counts = H.mg.apply_counts()
    # the same K cycles replayed as a hipGraph (one launch per cycle instead of ~120; single rank only)
    graph_ms = None
    if not multi or (args.graph_exchange and comm is not None):
        H.mg.set_graph(True)
        for _ in range(2):
            H.mg.apply(b, x)
        sync_all()
        tg = time.perf_counter()
        for _ in range(args.steps):
            H.mg.apply(b, x)
        sync_all()
        graph_ms = 1e3 * (time.perf_counter() - tg) / args.steps
        H.mg.set_graph(False)
    rn = H.mg.apply(b, x, verbose=True)
    log(f"[rank {rank}] residual norm after {args.warmup + args.steps + 1} cycles: {rn:.3e}; "
        f"stiffness launches per cycle (coarse->fine): {counts}")

    # ---- dominant kernel: the p = P stiffness kernel.  Timed where it runs: HIP events on the launch
    # stream around every run of stiffness launches of the fine operator during `prof_cycles` extra
    # V-cycles (7 applications of 8 colour launches each per cycle); the back-to-back replay of the same
    # launches outside a cycle is reported next to it (`replay_kernel_ms`).
    op = H.operators[-1]
    prof_cycles = 5
    op.set_profiling(True)
    for _ in range(prof_cycles):
        H.mg.apply(b, x)
    torch.cuda.synchronize()
    prof_ms, prof_launches = op.read_profile()
    op.set_profiling(False)
    kernel_ms = prof_ms / prof_launches
    u, y = H.new_vector(), H.new_vector()
    u.data.copy_(torch.randn(H.levels[-1].ndofs, dtype=torch.float64, device="cuda",
                             generator=torch.Generator(device="cuda").manual_seed(0)))
    op.time_kernel(u, y, 3)
    replay_ms = op.time_kernel(u, y, args.kernel_reps)
    nlaunch = op.launches_per_apply()  # one launch per patch colour (and cell list)
    ncells_launch = H.part.ncells / nlaunch  # mean cells per launch (owned + ghost layer, all colours)
    alg_bytes = algorithmic_bytes_per_cell(P) * ncells_launch
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    for tname in ("hbm_traffic_r04.json", "hbm_traffic_r03.json", "hbm_traffic_r02.json", "hbm_traffic_r01.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"stiffness_p{P}_bytes_per_launch")
                traffic_src = "profiles/" + tname
            except Exception:
                traffic = None
            if traffic is not None:
                break
    # in-run streaming ceiling of THIS box (boxes of the pool differ by a few per cent): the library's own
    # element-wise launcher on three fine-level vectors, r = alpha x + y (two reads + one write of 8 bytes per
    # dof; 3 x 136 MB at config 2, beyond the 256 MB Infinity Cache), HIP events on the launch stream
    tri = [H.new_vector() for _ in range(3)]
    for v in tri:
        v.set(1.0)
    for _ in range(3):
        pm.axpy(tri[0], 0.5, tri[1], tri[2])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tri_reps = 20
    e0.record()
    for _ in range(tri_reps):
        pm.axpy(tri[0], 0.5, tri[1], tri[2])
    e1.record()
    torch.cuda.synchronize()
    measured_peak = 24.0 * H.levels[-1].size_local / (e0.elapsed_time(e1) / tri_reps * 1e-3) / 1e9
    # ... and the runtime's device-to-device copy of one such vector ("hipMemcpyDtoD / triad ceiling")
    for _ in range(3):
        tri[0].data.copy_(tri[1].data)
    e0.record()
    for _ in range(tri_reps):
        tri[0].data.copy_(tri[1].data)
    e1.record()
    torch.cuda.synchronize()
    measured_copy = 16.0 * tri[0].data.numel() / (e0.elapsed_time(e1) / tri_reps * 1e-3) / 1e9
    del tri
    roofline = {"bound": "hbm", "kernel": f"stiffness_column_kernel<{P}>",
                "byte_model": "storedG (SURVEY.md 8d): 48N + 4N + 8 + 17U bytes per cell", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                # what a pure streaming kernel reaches on this box in this run (triad, see above), and the kernel
                # against that
                "measured_peak": round(measured_peak, 1), "frac_of_measured": round(achieved / measured_peak, 4),
                "measured_peak_kernel": "r = alpha x + y on three fine-level vectors (24 bytes per dof), "
                                        f"{tri_reps} launches, HIP events",
                "measured_copy_peak": round(measured_copy, 1),
                "traffic": traffic,
                "traffic_source": f"cached rocprofv3 PMC pass ({traffic_src}), not measured in this run",
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": round(kernel_ms, 5), "timing": f"HIP events around the {prof_launches} in-cycle launches "
                                                            f"of {prof_cycles} V-cycles",
                "replay_kernel_ms": round(replay_ms, 5),
                "replay_frac": round(alg_bytes / (replay_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "cells_per_launch": ncells_launch,
                "launches_per_apply": nlaunch, "apply_ms": round(kernel_ms * nlaunch, 5)}

    if not multi:
        exchange_note = "none (single rank)"
    elif comm is None:
        exchange_note = comm_note or "torch.distributed callbacks"
    elif args.comm == "windows":
        exchange_note = ("library communicator made of windows: halo and reductions as direct stores into IPC-mapped "
                         "device memory, no transport library")
    elif args.halo == "windows":
        exchange_note = ("library halo windows (direct stores into the neighbours' IPC-mapped memory), RCCL device "
                         "all-reduce")
    else:
        exchange_note = "library RCCL communicator (grouped send/recv, device all-reduce)"
    out = {
        "metric": "DoFs/sec per p-MG V-cycle (Poisson, hex, p=4)",
        "value": value,
        "unit": "DoF/s",
        "n_gpus": 1 if args.share_gpu else world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"Poisson p={P} V-cycle p=" + "->".join(str(p) for p in reversed(orders))
                        + f", Chebyshev({args.cheb})/Jacobi, {args.n}^3 hexes per GPU" if args.scaling == "weak"
                        else f"Poisson p={P} V-cycle, {args.n}^3 hexes total",
            "cells_global": list(n_global),
            "fine_dofs_global": fine_dofs_global,
            "levels": list(reversed(orders)),
            "cheb_iterations": args.cheb,
            "partition": "x".join(str(d) for d in dims) + " bricks, 1 ghost-cell layer",
            "exchange": exchange_note,
            "stiffness_launches_per_cycle": {f"p{p}": c for p, c in zip(orders, counts)},
            # ranks of the communicator the halo and the reductions really ran on (pmg_comm_size of the library's
            # RCCL communicator, or torch.distributed's RCCL group on the callback route); 1 = no communicator
            "rccl_ranks": rccl_ranks if args.comm == "rccl" else 0,
            # the same for whichever communicator carried the run (RCCL or the one made of windows)
            "comm_ranks": rccl_ranks,
            # a halo exchange captured into a hipGraph keeps its overlap with the interior cells only on a HIP >= 7.2
            # runtime (profiles/rccl_capture_probe_r03.md); a Python process runs on PyTorch's bundled runtime
            "captured_exchange_overlaps": bool(pm._lib.lib().pmg_comm_capture_overlaps()) if multi else None,
        },
        "roofline": roofline,
    }
    watch["out"] = out
    watch["phase"] = "after the headline"
    if args.share_gpu:
        out["rehearsal"] = (f"{world} ranks SHARE ONE GPU (--share-gpu): a rehearsal of the multi-rank path with real "
                            "inter-process exchanges, not a multi-GPU measurement; `value` is the aggregate of the "
                            "ranks time-slicing that GPU")
        out["ranks"] = world
    if graph_ms is not None:
        out["graph_replay"] = {"ms_per_step": graph_ms, "value": fine_dofs_global / (graph_ms * 1e-3), "unit": "DoF/s",
                               "note": "the same cycle replayed with one hipGraphLaunch per cycle"
                                       + (" (halo exchange captured on the compute stream)" if multi else "")
                                       + "; `value` above is the eager (stream-ordered launches) figure"}

    def timed_cycles(Hx, bx, xx, k):
        sync_all()
        t = time.perf_counter()
        for _ in range(k):
            Hx.mg.apply(bx, xx)
        sync_all()
        t = time.perf_counter() - t
        if multi:
            tt = torch.tensor([t], dtype=torch.float64, device=ctl)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    # ---- BASELINE config 3 (N > 1): the SAME 64^3 problem split over the N GPUs (strong scaling), next to the
    # weak-scaled headline -- and the run's NUMERIC GATE: rank 0 runs the single-domain C oracle on that problem
    # (the reference prints the residual per cycle of its `srun -n 8` run, examples/pmg/main.cpp:362-367,
    # examples/pmg/submit.sh:29; here every route has to reproduce the oracle's iterate and residual history before
    # it is timed).  A route that fails keeps its diagnostics and loses its numbers; if the route `value` ran on
    # fails -- or no route passes -- the whole line fails (value null, exit code 1).
    parity_failures = []

    def gather_global(vec, lv, nglobal):
        """The distributed vector on rank 0 in the single-domain numbering (owned entries, zero-padded sum)."""
        g = torch.zeros(nglobal, dtype=torch.float64, device=ctl)
        idx = torch.from_numpy(np.asarray(lv.local_to_global[: lv.size_local], dtype=np.int64)).to(ctl)
        g[idx] = vec.data[: lv.size_local].to(ctl)
        dist.reduce(g, dst=0, op=dist.ReduceOp.SUM)
        return g.cpu().numpy() if rank == 0 else None

    def _strong_block():
        res = {"workload": f"BASELINE config 3: {args.n}^3 hexes in total over {world} "
                           + ("ranks sharing one GPU" if args.share_gpu else "GPUs"),
               "scaling": "strong", "unit": "DoF/s", "routes": {}}
        oracle = {}
        value_route = None
        headline_route = None  # the route that runs on the headline's own communicator, eagerly, as `value` was timed
        route_failures = {}
        for name, ckind, halo, captured, skip in strong_routes(world, args.share_gpu, args.comm):
            if skip:
                res["routes"][name] = {"skipped": skip}
                continue
            entry = {"communicator": "RCCL" if ckind == "rccl" else "windows (direct stores, no transport library)",
                     "halo": halo, "captured": captured}
            watch["phase"] = f"strong-scaling route {name}"
            try:
                if ckind == args.comm and halo == (args.halo if ckind == "rccl" else "windows") and comm is not None:
                    rcomm = comm  # the communicator of the headline
                elif ckind == "windows":
                    rcomm = pm.WindowComm.from_torch()
                else:
                    rcomm = pm.RcclComm.from_torch(device=torch.device("cuda", local_rank), halo=halo)
                made = 1
            except Exception as e:  # every rank must take the same branch: agree below
                log(f"[rank {rank}] route {name}: communicator unavailable ({type(e).__name__}: {e})")
                rcomm, made = None, 0
            okc = torch.tensor([made], device=ctl)
            dist.all_reduce(okc, op=dist.ReduceOp.MIN)
            if int(okc.item()) == 0:
                res["routes"][name] = {**entry, "skipped": "communicator could not be created on every rank"}
                continue
            if rcomm is comm and (headline_route is None or (not captured and res["routes"][headline_route]["captured"])):
                headline_route = name  # (the eager route on that communicator if there is one, else the captured one)

            def hook(lv):
                if args.corrupt_halo and rank == world - 1 and lv.P == P and len(lv.recv_indices) >= 8:
                    # the middle half of the ghost slots in reverse order (the first slots of a list are Dirichlet
                    # dofs of the domain boundary, whose values never enter the operator)
                    ng_ = len(lv.recv_indices)
                    lv.recv_indices = lv.recv_indices.copy()
                    lv.recv_indices[ng_ // 4: 3 * ng_ // 4] = lv.recv_indices[ng_ // 4: 3 * ng_ // 4][::-1].copy()

            Hs = pm.PoissonHierarchy((args.n,) * 3, orders, kappa=2.0, cheb_its=args.cheb, proc_dims=dims, rank=rank,
                                     size=world, comm=rcomm, level_hook=hook)
            lvs = Hs.levels[-1]
            nd_s = Hs.part.global_ndofs(P)
            bs_ = Hs.rhs[-1]
            if not oracle:  # once: the single-domain C oracle on rank 0, with the distributed run's smoother bounds
                bg = gather_global(bs_, lvs, nd_s)
                if rank == 0:
                    from oracle import c_oracle as co

                    t0 = time.time()
                    ncores = co.cpu_share()
                    co.set_num_threads(ncores)
                    part0 = pm.BoxPartition((args.n,) * 3)
                    assert np.array_equal(part0.level(P).local_to_global, np.arange(nd_s))
                    cl = [co.CLevel(p_, 2.0, part0.level(p_).dofmap, part0.xgeom, part0.geom_dofmap,
                                    part0.level(p_).bc_marker) for p_ in orders]
                    ci = [co.CInterp(cl[i], cl[i + 1]) for i in range(len(orders) - 1)]
                    cm = co.CMultigrid(cl, ci, [e[1] for e in Hs.eig_ranges], args.cheb)
                    xc = np.zeros_like(bg)
                    rns_o, t_cyc = [], []
                    for c in range(GATE_CYCLES + 1):
                        if c == GATE_CYCLES:
                            oracle["x"] = xc.copy()
                        tc = time.perf_counter()
                        cm.apply(bg, xc)
                        t_cyc.append(time.perf_counter() - tc)
                        if c < GATE_CYCLES:
                            rns_o.append(float(np.linalg.norm(bg - cl[-1].apply(xc))))
                    cpu_s = float(np.mean(t_cyc[1:]))
                    oracle.update(rns=rns_o, cpu_s=cpu_s, cores=co.num_threads())
                    log(f"gate oracle: setup + {GATE_CYCLES + 1} cycles {time.time() - t0:.1f}s, {cpu_s:.2f}s per V-cycle "
                        f"on {co.num_threads()} threads")
                    out["cpu_baseline"] = {
                        "value": nd_s / cpu_s, "unit": "DoF/s", "cores": co.num_threads(), "kind": "port",
                        "algorithmic_GBs": round(oracle_cycle_bytes(orders, part0.ncells, [
                            part0.level(p_).ndofs for p_ in orders], args.cheb) / cpu_s / 1e9, 1),
                        "host_logical_cpus": os.cpu_count(),
                        "sample": f"{GATE_CYCLES} V-cycles of the {args.n}^3-hex problem ({nd_s} fine dofs: BASELINE "
                                  f"config 2 = the strong-scaling problem of this run; the weak-scaled headline is "
                                  f"{world} such bricks), C/OpenMP oracle on rank 0's CPU quota, after 1 warm-up cycle"}
                    del cl, ci, cm, part0
                else:
                    oracle["x"] = None
            # the route's first GATE_CYCLES cycles from x0 = 0, as it will be timed (eager or replayed)
            xs_ = Hs.new_vector()
            xs_.set(0.0)
            Hs.mg.set_graph(bool(captured))  # explicit: the library's default may be either
            rns_g = [Hs.mg.apply(bs_, xs_, verbose=True) for _ in range(GATE_CYCLES)]
            xg = gather_global(xs_, lvs, nd_s)
            passed = torch.tensor([0], device=ctl)
            if rank == 0:
                ok, rec = gate_verdict(xg, oracle["x"], rns_g, oracle["rns"])
                entry["gate"] = rec
                passed[0] = 1 if ok else 0
            dist.broadcast(passed, src=0)
            if int(passed.item()) == 1:
                for _ in range(max(args.warmup - GATE_CYCLES, 1)):
                    Hs.mg.apply(bs_, xs_)
                ts = timed_cycles(Hs, bs_, xs_, args.steps)
                entry.update(value=nd_s * args.steps / ts, ms_per_step=1e3 * ts / args.steps)
                if captured:
                    entry["graph_replays"] = Hs.mg.graph_replays()
                    entry["captured_exchange_overlaps"] = (bool(pm._lib.lib().pmg_comm_capture_overlaps())
                                                           if ckind == "rccl" and halo == "exchange" else True)
                if value_route is None:
                    value_route = name
            else:
                entry.update(value=None, parity_failed=True)
                route_failures[name] = (f"strong-scaling route {name}: the {world}-rank cycle does not reproduce the "
                                        f"single-domain C oracle ({entry.get('gate')})")
            Hs.mg.set_graph(None)
            res["routes"][name] = entry
            res.update(fine_dofs_global=nd_s, local_dofs=[lv.size_local for lv in Hs.levels],
                       ghosts=[lv.num_ghosts for lv in Hs.levels])
            del Hs, xs_, bs_
            if rcomm is not comm:
                del rcomm
            import gc

            gc.collect()
            torch.cuda.synchronize()
            dist.barrier()
        if rank == 0:
            # The headline ran on ONE transport.  It is void if that transport's route fails the gate, if no route
            # passes, or if no route ran on its communicator at all; another route that fails loses its own numbers and
            # is named in the line (`parity.routes_failed`), loudly, without taking a validated headline with it.
            if value_route is None:
                parity_failures.append("strong-scaling gate: no route reproduced the C oracle")
                parity_failures.extend(route_failures.values())
            elif headline_route is None or headline_route in route_failures:
                parity_failures.extend(route_failures.values())
                if headline_route is None:
                    parity_failures.append("strong-scaling gate: no route ran on the headline's communicator")
            for msg in route_failures.values():
                log("GATE FAILURE: " + msg)
            if value_route is None:
                res.update(value=None, ms_per_step=None, value_route=None)
            else:
                vr = res["routes"][value_route]
                res.update(value=vr["value"], ms_per_step=vr["ms_per_step"], value_route=value_route)
            out["strong_scaling"] = res
            out["parity"] = {"gate": f"every timed multi-rank route reproduces {GATE_CYCLES} V-cycles of the single-domain "
                                     f"C oracle on the {args.n}^3 problem (iterate {GATE_TOLERANCE:g}, residual norms "
                                     f"{GATE_RNORM_TOLERANCE:g}); the weak-scaled `value` runs the same kernels and "
                                     f"the same exchange on {world} such bricks",
                             "routes_passed": [k for k, v in res["routes"].items() if v.get("gate", {}).get("passed")],
                             "routes_failed": sorted(route_failures),
                             "headline_route": headline_route}
        # all ranks agree on failure (rank 0 decides)
        nf = torch.tensor([len(parity_failures) if rank == 0 else 0], device=ctl)
        dist.broadcast(nf, src=0)
        if rank != 0 and int(nf.item()) > 0:
            parity_failures.append("gate failed on rank 0")

    if multi and args.scaling == "weak" and not args.no_strong:
        _strong_block()
        watch["phase"] = "after the strong-scaling block"
    elif multi:
        out["parity"] = {"gate": "unverified: the strong-scaling block and its gate against the C oracle were skipped "
                                 "(--no-strong or --scaling strong)"}

    # ---- the cycle with its coarsest level SOLVED (the reference's --amg, examples/pmg/main.cpp:331-335): the
    # library's AMG on the degree-1 level, as CG <= 60 iterations / rtol 1e-5 (the reference's shape) and as
    # two stationary AMG cycles (no host synchronisation).  Reported next to the headline, which keeps the
    # reference's default (coarsest level = its Chebyshev smoother, src/pmg.hpp:108-109).  Not `value`.
    def _contraction(k=6):
        xc_ = H.new_vector()
        xc_.set(0.0)
        rns = [H.mg.apply(b, xc_, verbose=True) for _ in range(k)]
        return [rns[i + 1] / rns[i] for i in range(k - 1)], xc_

    def make_amg(**kw):
        """The coarse solver: on several ranks the replicated hierarchy (global degree-1 matrix on every rank,
        one all-reduce per solve), on one rank the plain one."""
        if multi:  # first coarsening per rank, only level 1 gathered: no rank holds the global degree-1 matrix
            return pm.AmgSolver(H.operators[0], global_index=H.levels[0].local_to_global,
                                n_global=H.part.global_ndofs(orders[0]), setup="distributed", **kw)
        return pm.AmgSolver(H.operators[0], **kw)

    def _amg_coarse():
        res = {"plain_cycle": {"coarse": "Chebyshev smoother (reference default)", "ms_per_step": ms_per_step,
                               "residual_contraction_per_cycle": [round(c, 4) for c in _contraction()[0]]}}
        t_setup = time.perf_counter()
        modes = [("krylov", dict(max_iter=60, rtol=1e-5)), ("stationary_2_cycles", dict(cycles=2))]
        for name, kw in modes:
            amg = make_amg(**kw)
            torch.cuda.synchronize()
            setup_s = time.perf_counter() - t_setup
            H.mg.set_coarse_solver(amg)
            contr, xa = _contraction()
            for _ in range(2):
                H.mg.apply(b, xa)
            ta = timed_cycles(H, b, xa, args.steps)
            res[name] = {"ms_per_step": 1e3 * ta / args.steps, "value": fine_dofs_global * args.steps / ta,
                         "unit": "DoF/s", "coarse_share_of_cycle": round(1.0 - ms_per_step / (1e3 * ta / args.steps), 4),
                         "residual_contraction_per_cycle": [round(c, 4) for c in contr],
                         "amg_levels": amg.info(), "setup_seconds": round(setup_s, 3)}
            H.mg.set_coarse_solver(None)
            t_setup = time.perf_counter()
            del amg, xa
        out["amg_coarse"] = res

    # ---- BASELINE config 2 as worded: CG preconditioned by the V-cycle, to rtol 1e-8 (extra, not `value`).
    # Two right-hand sides: a seeded random one (the honest measure of the preconditioner: every mode is
    # excited) and the manufactured one (an eigenfunction of the continuous operator, which CG resolves
    # regardless of the preconditioner's quality).
    def _pcg():
        res = {"preconditioner": "V-cycle, zero initial guess", "rtol": 1e-8}
        lvf = H.levels[-1]
        brand = H.new_vector()
        g = np.random.default_rng(1000 + rank).standard_normal(lvf.ndofs)
        g[lvf.bc_marker.astype(bool)] = 0.0
        brand.data.copy_(torch.from_numpy(g))
        amg = make_amg(cycles=2)
        for name, rhs in (("random_rhs", brand), ("manufactured_rhs", b), ("random_rhs_amg_coarse", brand)):
            cg = pm.CGSolver(H.layouts[-1])
            cg.set_max_iterations(200)
            cg.set_tolerance(1e-8)
            if name.endswith("amg_coarse"):
                H.mg.set_coarse_solver(amg)
                cg.set_flexible(False)  # stationary AMG cycles: the V-cycle stays a fixed linear operator
            xs = H.new_vector()
            xs.set(0.0)
            sync_all()
            t_pcg = time.perf_counter()
            pcg_its = cg.solve(H.operators[-1], xs, rhs, preconditioner=H.mg)
            sync_all()
            t_pcg = time.perf_counter() - t_pcg
            rr = H.new_vector()
            H.operators[-1](xs, rr)
            pm.axpy(rr, -1.0, rr, rhs)
            res[name] = {"iterations": pcg_its, "seconds": t_pcg,
                         "true_relative_residual": pm.norm(rr) / pm.norm(rhs)}
            H.mg.set_coarse_solver(None)
            del cg, xs, rr
        out["pcg"] = res

    # ---- BASELINE config 5: p = 6 -> 3 -> 1 with the AMG coarse solve at p = 1 inside the V-cycle (the reference:
    # hypre BoomerAMG, src/amg.hpp; here csrc/amg.hip, two stationary cycles), ~17 M dofs per GPU (43^3 hexes).  Extra,
    # not `value`; its parity is the small-mesh tests' (tests/test_gpu_distributed.py, orders (1, 3, 6)).
    def _config5():
        n5 = 43
        dims5 = dims
        H5 = pm.PoissonHierarchy(tuple(n5 * d for d in dims5), (1, 3, 6), kappa=2.0, cheb_its=args.cheb, proc_dims=dims5,
                                 rank=rank, size=world, comm=comm)
        t_setup = time.perf_counter()
        if multi:
            amg5 = pm.AmgSolver(H5.operators[0], global_index=H5.levels[0].local_to_global,
                                n_global=H5.part.global_ndofs(1), cycles=2, setup="distributed")
        else:
            amg5 = pm.AmgSolver(H5.operators[0], cycles=2)
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t_setup
        H5.mg.set_coarse_solver(amg5)
        x5 = H5.new_vector()
        x5.set(0.0)
        rns = [H5.mg.apply(H5.rhs[-1], x5, verbose=True) for _ in range(6)]
        for _ in range(2):
            H5.mg.apply(H5.rhs[-1], x5)
        t5 = timed_cycles(H5, H5.rhs[-1], x5, args.steps)
        nd5 = H5.part.global_ndofs(6)
        out["config5"] = {"workload": f"Poisson p=6 V-cycle p=6->3->1, Chebyshev({args.cheb})/Jacobi, AMG (2 stationary "
                                      f"cycles) on the p=1 level, {n5}^3 hexes per GPU",
                          "value": nd5 * args.steps / t5, "unit": "DoF/s", "ms_per_step": 1e3 * t5 / args.steps,
                          "fine_dofs_global": nd5,
                          "residual_contraction_per_cycle": [round(rns[i + 1] / rns[i], 4) for i in range(5)],
                          "amg_levels": amg5.info(), "amg_setup_seconds": round(setup_s, 3)}
        H5.mg.set_coarse_solver(None)
        del amg5, x5, H5

    extras = world == 1 or args.extras  # the scaling runs measure the headline only
    if extras:
        with extra(out, "amg_coarse"):
            _amg_coarse()
        with extra(out, "pcg"):
            _pcg()
        if not args.no_config5:
            with extra(out, "config5"):
                _config5()

    # ---- h-independence of the outer PCG with a random right-hand side (extra, --mesh-sweep, N = 1) ----
    def _mesh_sweep():
        res = {}
        for ns in (32, 64, 96):
            Hm = H if ns == args.n else pm.PoissonHierarchy(ns, orders, kappa=2.0, cheb_its=args.cheb)
            lvm = Hm.levels[-1]
            g = np.random.default_rng(7).standard_normal(lvm.ndofs)
            g[lvm.bc_marker.astype(bool)] = 0.0
            bm = Hm.new_vector()
            bm.data.copy_(torch.from_numpy(g))
            entry = {}
            for coarse in ("chebyshev", "amg_2_cycles"):
                amg = pm.AmgSolver(Hm.operators[0], cycles=2) if coarse != "chebyshev" else None
                Hm.mg.set_coarse_solver(amg)
                cg = pm.CGSolver(Hm.layouts[-1])
                cg.set_max_iterations(200)
                cg.set_tolerance(1e-8)
                xm = Hm.new_vector()
                xm.set(0.0)
                torch.cuda.synchronize()
                tm = time.perf_counter()
                its = cg.solve(Hm.operators[-1], xm, bm, preconditioner=Hm.mg)
                torch.cuda.synchronize()
                entry[coarse] = {"iterations": its, "seconds": round(time.perf_counter() - tm, 4)}
                Hm.mg.set_coarse_solver(None)
                del cg, xm, amg
            res[f"{ns}^3"] = entry
            if Hm is not H:
                del Hm
            torch.cuda.empty_cache()
        out["pcg_mesh_sweep"] = {"rtol": 1e-8, "rhs": "standard normal, seed 7", **res}

    if world == 1 and args.mesh_sweep:
        with extra(out, "pcg_mesh_sweep"):
            _mesh_sweep()

    # ---- extra, reported separately and never mixed into `value`/`roofline`: the same V-cycle with the
    # affine-cell geometry mode (one constant tensor per cell instead of the stored G stream; byte model
    # "cellG" = 4N + 56 + 17U bytes per cell).  Every cell of a box mesh is a parallelepiped.
    def _affine_geometry():
        if not args.no_affine and all(o.is_affine() for o in H.operators):
            for o in H.operators:
                o.set_geometry_mode("affine")
            try:
                _affine_body()
            finally:
                for o in H.operators:
                    o.set_geometry_mode("stored")

    def _affine_body():
        if True:
            xa = H.new_vector()
            xa.set(0.0)
            for _ in range(args.warmup + 1):  # x above has seen warmup + steps + 1 cycles
                H.mg.apply(b, xa)
            sync_all()
            ta = time.perf_counter()
            for _ in range(args.steps):
                H.mg.apply(b, xa)
            sync_all()
            ta = time.perf_counter() - ta
            if world > 1:
                tt = torch.tensor([ta], dtype=torch.float64, device=ctl)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ta = float(tt.item())
            op.time_kernel(u, y, 3)
            kms = op.time_kernel(u, y, args.kernel_reps)
            N_, U_ = (P + 1) ** 3, P**3
            cellg = (4 * N_ + 56 + 17 * U_) * ncells_launch
            # same iterates as the stored-G cycle?  (three cycles from x = 0 in either mode)
            xa.set(0.0)
            for _ in range(3):
                H.mg.apply(b, xa)
            for o in H.operators:
                o.set_geometry_mode("stored")
            xs3 = H.new_vector()
            xs3.set(0.0)
            for _ in range(3):
                H.mg.apply(b, xs3)
            nl = H.levels[-1].size_local
            out["affine_geometry"] = {
                "value": fine_dofs_global * args.steps / ta, "unit": "DoF/s", "ms_per_step": 1e3 * ta / args.steps,
                "note": "same V-cycle, geometry mode 'affine' (not the reference's data structure); not `value`",
                "roofline": {"bound": "hbm", "byte_model": "cellG: 4N + 56 + 17U bytes per cell",
                             "achieved": round(cellg / (kms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(cellg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms": round(kms, 5)},
                "max_rel_diff_vs_stored_after_3_cycles": float((xa.data[:nl] - xs3.data[:nl]).abs().max()
                                                               / xs3.data[:nl].abs().max()),
            }
            del xs3
            del xa

    if extras:
        with extra(out, "affine_geometry"):
            _affine_geometry()

    # ---- BASELINE config 4 (extra, N = 1): operator apply alone for p in {2, 4, 6, 8} at ~17 M dofs, same
    # byte model and timing hook as `roofline` ----

    def _degree_sweep():
        if world == 1 and not args.no_sweep:
            sweep = {}
            for Ps, ns in ((2, 128), (4, 64), (6, 43), (8, 32)):
                parts = pm.BoxPartition(ns)
                lvs = parts.level(Ps)
                lays = pm.make_layout(lvs)
                ops = pm.MatFreeLaplacian(Ps, 2.0, lvs.dofmap, parts.xgeom, parts.geom_dofmap, lvs.lcells, lvs.bcells,
                                          lvs.bc_marker, lays)
                us, ys = pm.Vector(lays), pm.Vector(lays)
                uh = np.random.default_rng(Ps).standard_normal(lvs.ndofs)
                us.data.copy_(torch.from_numpy(uh))
                # parity of exactly the launches that are timed (the coloured path) against the C oracle
                ops(us, ys)
                torch.cuda.synchronize()
                entry = {"cells": ns**3, "dofs": lvs.ndofs}
                if not args.no_cpu:
                    from oracle import c_oracle as co

                    co.set_num_threads(co.cpu_share())
                    cl = co.CLevel(Ps, 2.0, lvs.dofmap, parts.xgeom, parts.geom_dofmap, lvs.bc_marker)
                    ref = cl.apply(uh)
                    err = float(np.abs(ys.data_copy() - ref).max() / np.abs(ref).max())
                    entry["rel_err_vs_oracle"] = err
                    del cl, ref
                    if not err < 1e-12:
                        parity_failures.append(f"degree_sweep p{Ps}: rel. err {err:.3e} vs the C oracle")
                        sweep[f"p{Ps}"] = {**entry, "error": "parity failure, timing withheld"}
                        del ops, us, ys, lays, lvs, parts
                        torch.cuda.empty_cache()
                        continue
                ops.time_kernel(us, ys, 3)
                ms = ops.time_kernel(us, ys, args.kernel_reps) * ops.launches_per_apply()
                gbs = algorithmic_bytes_per_cell(Ps) * parts.ncells / (ms * 1e-3) / 1e9
                sweep[f"p{Ps}"] = {**entry, "launches": ops.launches_per_apply(), "apply_ms": round(ms, 5),
                                   "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
                if Ps == 4:
                    # the same application in its chain form (include/pmg_amd.h "Chain form"; off by default: measured
                    # slower, profiles/kernel_tuning_r04.md section 9), checked against the patch launches' vector
                    keep = os.environ.get("PMG_CHAIN")
                    os.environ["PMG_CHAIN"] = "1"
                    try:
                        opc = pm.MatFreeLaplacian(Ps, 2.0, lvs.dofmap, parts.xgeom, parts.geom_dofmap, lvs.lcells,
                                                  lvs.bcells, lvs.bc_marker, lays)
                    finally:
                        if keep is None:
                            del os.environ["PMG_CHAIN"]
                        else:
                            os.environ["PMG_CHAIN"] = keep
                    if opc.chain_available():
                        yc = pm.Vector(lays)
                        opc(us, yc)
                        torch.cuda.synchronize()
                        errc = float((yc.data - ys.data).abs().max() / ys.data.abs().max())
                        opc.time_kernel(us, yc, 3)
                        msc = opc.time_kernel(us, yc, args.kernel_reps) * opc.launches_per_apply()
                        gbc = algorithmic_bytes_per_cell(Ps) * parts.ncells / (msc * 1e-3) / 1e9
                        sweep["p4"]["chain_form"] = {
                            "launches": opc.launches_per_apply(), "apply_ms": round(msc, 5), "achieved": round(gbc, 1),
                            "frac": round(gbc / HBM_PEAK_GBS, 4), "rel_diff_vs_patch_launches": errc,
                            "note": "one persistent workgroup per chain of patches; opt-in (PMG_CHAIN=1), not `value`"}
                        if not errc < 1e-12:
                            parity_failures.append(f"degree_sweep p4 chain form: rel. diff {errc:.3e}")
                        del yc
                    del opc
                del ops, us, ys, lays, lvs, parts
                torch.cuda.empty_cache()
            out["degree_sweep"] = {"note": "operator apply only (back-to-back replays of the colour launches), model storedG, "
                                           "GB/s of 8000 (BASELINE config 4); every timed operator is first checked "
                                           "against the C oracle on the same vector (1e-12); not `value`",
                                   **sweep}

    with extra(out, "degree_sweep"):
        _degree_sweep()

    # ---- CPU baseline: the C/OpenMP port of the same lean V-cycle on the host cores (rank 0, N = 1) ----
    def _cpu_baseline():
        if world == 1 and not args.no_cpu:
            from oracle import c_oracle as co

            t0 = time.time()
            # threads = the cores this process may really use (affinity mask capped by the cgroup CPU quota): the GPU
            # box reports 128-256 logical CPUs but grants a 16-CPU quota, and 128 OpenMP threads under that quota run
            # 3x slower than 16 (gpurun_out/r3_cpu_probe.txt)
            ncores = co.cpu_share()
            co.set_num_threads(ncores)
            part = H.part
            cl = [co.CLevel(p, 2.0, part.level(p).dofmap, part.xgeom, part.geom_dofmap, part.level(p).bc_marker)
                  for p in orders]
            ci = [co.CInterp(cl[i], cl[i + 1]) for i in range(len(orders) - 1)]
            cm = co.CMultigrid(cl, ci, [e[1] for e in H.eig_ranges], args.cheb)
            bh = b.data_copy()
            xc = np.zeros_like(bh)
            cm.apply(bh, xc)  # first cycle from x0 = 0: also the parity check below
            ncpu = 4
            tc = time.perf_counter()
            for _ in range(ncpu):
                cm.apply(bh, xc)
            cpu_s = (time.perf_counter() - tc) / ncpu
            log(f"cpu baseline: setup {time.time() - t0 - cpu_s * ncpu:.1f}s, {cpu_s:.2f}s per V-cycle on "
                f"{co.num_threads()} threads")
            # parity in the same run: 1 + ncpu GPU V-cycles from x0 = 0 against the CPU ones
            xg = H.new_vector()
            xg.set(0.0)
            for _ in range(1 + ncpu):
                H.mg.apply(b, xg)
            torch.cuda.synchronize()
            got = xg.data_copy()
            err = float(np.abs(got - xc).max() / np.abs(xc).max())
            out["cpu_baseline"] = {"value": fine_dofs_global / cpu_s, "unit": "DoF/s", "cores": co.num_threads(),
                                   "kind": "port",
                                   # bytes of the cycle the oracle REALLY runs (7 / 6 / 2 applications at k = 3), not of
                                   # the survey's lean-cycle definition (9 / 9 / 4)
                                   "algorithmic_GBs": round(oracle_cycle_bytes(orders, part.ncells, [
                                       part.level(p).ndofs for p in orders], args.cheb) / cpu_s / 1e9, 1),
                                   "host_logical_cpus": os.cpu_count(),
                                   "sample": f"{ncpu} V-cycles of the same workload ({args.n}^3 hexes, "
                                             f"{fine_dofs_global} fine dofs), C/OpenMP oracle (cell-coloured scatter, no "
                                             f"atomics) on the process's CPU quota, after 1 warm-up cycle"}
            out.setdefault("parity", {}).update({f"gpu_vs_cpu_oracle_rel_err_after_{1 + ncpu}_cycles": err,
                                                 "tolerance": 1e-10})
            if not err < 1e-10:
                parity_failures.append(f"V-cycle: rel. err {err:.3e} vs the C oracle after {1 + ncpu} cycles")

    # the parity check is not an "extra": a failure (or an exception in it) fails the run
    _cpu_baseline()
    for f in parity_failures:
        log("PARITY FAILURE: " + f)
    finish_line(out, parity_failures)

    if deadline > 0:
        timer.cancel()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    # release the library objects (and with them the RCCL communicator) while the runtime is still up
    del H, x, b, u, y, op
    comm = None
    import gc

    gc.collect()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failures:
        sys.exit(1)


if __name__ == "__main__":
    main()
