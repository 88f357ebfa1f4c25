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
