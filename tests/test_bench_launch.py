"""bench.py's launch contract (the reference's run is `srun -n 8`, examples/pmg/submit.sh:29):
`--gpus N` never yields a line for another number of ranks -- it starts its N ranks itself as child processes,
or exits non-zero with a message.  CPU only: no GPU is touched."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(kw)
    return env


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _no_gpu_here():
    import torch

    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu_here(), reason="the refusal for missing devices is checked on the GPU-less box")
def test_gpus_2_without_devices_is_refused_not_downgraded():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""  # no JSON line of a smaller run
    assert "--gpus 2" in r.stderr and "refusing" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "WORLD_SIZE=4" in r.stderr


def test_self_launch_starts_the_ranks_as_children(monkeypatch):
    """With enough devices and no launcher in the environment, `--gpus N` becomes N child processes
    `python bench.py <same arguments>` with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, rendezvous on
    127.0.0.1 and a free port); the parent never initialises a GPU and returns the first non-zero exit code."""
    b = _bench_module()
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None):
            self.cmd, self.env = cmd, env
            self.returncode = 7 if env["RANK"] == "2" else 0
            started.append(self)

        def poll(self):
            return self.returncode

        def wait(self, timeout=None):
            return self.returncode

        def terminate(self):
            pass

    monkeypatch.setattr(b, "visible_gpus", lambda: 4)
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", [BENCH, "--gpus", "4", "--steps", "2", "--n", "32"])
    args = b.parse_args(["--gpus", "4", "--steps", "2", "--n", "32"])
    assert b.self_launch(args) == 7
    assert len(started) == 4
    ports = set()
    for r, p in enumerate(started):
        assert p.cmd[0] == sys.executable and os.path.samefile(p.cmd[1], BENCH)
        assert p.cmd[2:] == ["--gpus", "4", "--steps", "2", "--n", "32"]  # the script's own options, untouched
        assert (p.env["RANK"], p.env["LOCAL_RANK"], p.env["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert p.env["MASTER_ADDR"] == "127.0.0.1" and p.env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
        ports.add(int(p.env["MASTER_PORT"]))
    assert len(ports) == 1 and 1024 < ports.pop() < 65536


def test_self_launch_refuses_when_devices_are_short(monkeypatch):
    b = _bench_module()
    monkeypatch.setattr(b, "visible_gpus", lambda: 1)
    with pytest.raises(SystemExit) as e:
        b.self_launch(b.parse_args(["--gpus", "2"]))
    assert e.value.code == 2


def test_a_rank_that_fails_first_takes_the_others_down(monkeypatch):
    """`any(p.poll() is None ...)` stopped polling at the first child still running, so a rank k > 0 that
    died while rank 0 was alive went unseen until rank 0 left on its own (an RCCL / gloo timeout later).  Every child
    is polled every pass now: rank 2 fails on its second poll while ranks 0, 1, 3 keep running -- they are terminated
    and the launcher returns rank 2's code."""
    b = _bench_module()
    procs = []

    class FakeProc:
        def __init__(self, cmd, env=None):
            self.rank = int(env["RANK"])
            self.polls, self.returncode, self.terminated = 0, None, False
            procs.append(self)

        def poll(self):
            self.polls += 1
            if self.rank == 2 and self.polls >= 2:
                self.returncode = 9
            return self.returncode

        def wait(self, timeout=None):
            if self.returncode is None:
                self.returncode = -15 if self.terminated else 0
            return self.returncode

        def terminate(self):
            self.terminated = True
            self.returncode = -15

    monkeypatch.setattr(b, "visible_gpus", lambda: 4)
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(b.time, "sleep", lambda s: None)
    monkeypatch.setattr(sys, "argv", [BENCH, "--gpus", "4"])
    assert b.self_launch(b.parse_args(["--gpus", "4"])) == 9
    assert [p.terminated for p in procs] == [True, True, False, True]
    assert procs[0].polls <= 4  # seen on the pass it happened, not after rank 0 had left


def test_the_gate_refuses_a_wrong_multi_rank_cycle():
    """an N-rank line must prove its numbers.  The gate compares the N-rank iterate and residual
    history with the single-domain C oracle; a mismatch takes the route's numbers away, and if it is the route
    `value` came from (or no route passes) the line loses its `value` and the run exits 1."""
    import numpy as np

    b = _bench_module()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1000)
    rn = [3.0, 1.0, 0.4]
    ok, rec = b.gate_verdict(x * (1 + 1e-13), x, [r * (1 + 1e-10) for r in rn], rn)
    assert ok and rec["passed"] and rec["iterate_rel_err"] < 1e-12
    y = x.copy()
    y[17], y[18] = y[18], y[17]  # two ghost values in each other's slots somewhere upstream
    ok, rec = b.gate_verdict(y, x, rn, rn)
    assert not ok and not rec["passed"] and rec["iterate_rel_err"] > 1e-3
    ok, rec = b.gate_verdict(x, x, [3.0, 1.0, 0.41], rn)  # right iterate, wrong residual history
    assert not ok
    ok, rec = b.gate_verdict(np.full(1000, np.nan), x, rn, rn)
    assert not ok and "not finite" in rec["reason"]
    ok, rec = b.gate_verdict(x[:10], x, rn, rn)
    assert not ok
    # what the line looks like afterwards
    out = {"value": 2.5e10, "strong_scaling": {"value": None}}
    assert b.finish_line(out, ["strong-scaling route rccl_eager: ..."]) == 1
    assert out["value"] is None and out["parity_failed"] and out["parity"]["failures"]
    out = {"value": 2.5e10}
    assert b.finish_line(out, []) == 0 and out["value"] == 2.5e10 and "parity_failed" not in out


def test_strong_scaling_routes():
    """RCCL is the default route and the one `strong_scaling.value` comes from (it runs first); halo windows are
    measured next to it; ranks that share a GPU cannot use RCCL, their entries say so."""
    b = _bench_module()
    r = b.strong_routes(8, False, "rccl")
    assert [x[0] for x in r] == ["rccl_eager", "rccl_graph", "windows_graph"] and all(x[4] is None for x in r)
    assert r[2][1:4] == ("rccl", "windows", True)
    r = b.strong_routes(4, True, "windows")
    assert [x[0] for x in r] == ["rccl_eager", "rccl_graph", "windows_eager", "windows_graph"]
    assert "RCCL refuses" in r[0][4] and "RCCL refuses" in r[1][4] and r[2][4] is None and r[3][4] is None


def test_oracle_cycle_bytes_counts_the_applications_the_oracle_runs():
    """cpu_baseline.algorithmic_GBs uses the cycle oracle/pmg_oracle.c really runs: 7 / 6 / 2 operator applications
    at k = 3 (its source: orc_vcycle, orc_cheb_solve), not the survey's lean-cycle definition (9 / 9 / 4)."""
    b = _bench_module()
    orders, ncells, nd, k = (1, 2, 4), 64**3, [65**3, 129**3, 257**3], 3
    run, lean = b.oracle_cycle_bytes(orders, ncells, nd, k), b.cycle_algorithmic_bytes(orders, ncells, nd, k)
    assert abs(lean / 1e9 - 32.8) < 0.3
    assert 0.78 < run / lean < 0.86  # 27.1 GB against 32.9 GB
    cell = b.algorithmic_bytes_per_cell
    only_applies = b.oracle_cycle_bytes(orders, ncells, [0, 0, 0], k)
    maps = sum(2 * 4.0 * ((p + 1) ** 3 + (q + 1) ** 3) * ncells for p, q in ((2, 1), (4, 2)))
    assert abs(only_applies - maps - ncells * (7 * cell(4) + 6 * cell(2) + 2 * cell(1))) < 1.0


def test_a_run_that_stops_still_prints_its_line():
    """--deadline: a run that has not finished in time (on several ranks: a rank died and the others wait in a
    collective) prints ONE line without a `value`, naming the phase it was in, and exits with code 5.  Here the
    deadline falls into the start-up of a one-rank run (importing torch takes longer than 50 ms)."""
    import json

    r = subprocess.run([sys.executable, BENCH, "--deadline", "0.05", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 5, (r.returncode, r.stderr[-400:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and line["incomplete"] is True
    assert "no result after" in line["error"] and "start-up" in line["error"]
    assert line["metric"].startswith("DoFs/sec") and line["n_gpus"] == 1
