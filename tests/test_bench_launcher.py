"""bench.py's launcher logic on the CPU (no GPU, no processes started): `python bench.py --gpus N` without a launcher must re-exec
itself under torch.distributed.run with N ranks on 127.0.0.1 and pass its own arguments through; with fewer visible devices than
ranks it must exit non-zero instead (the one-device smoke flag aside).  The GPU side of the same contract — two real ranks, the
`n_gpus` / `ranks_seen` fields of the line — is tests/test_gpu_two_ranks_one_device.py."""
import importlib
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    sys.path.insert(0, ROOT)
    mod = importlib.import_module("bench")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    return mod


def _args(n):
    return types.SimpleNamespace(gpus=n)


def test_self_launch_builds_the_documented_command(bench, monkeypatch):
    import subprocess
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("PFK_BENCH_SHARED_DEVICE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(4))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    # torchrun picks the free port itself (`--standalone`), on 127.0.0.1: no probe socket of ours, no window for a port race
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]           # its own arguments, unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_self_launch_refuses_more_ranks_than_devices(bench, monkeypatch):
    import subprocess
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.delenv("PFK_BENCH_SHARED_DEVICE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(8))
    assert e.value.code not in (0, None) and "only 1 GPU" in str(e.value.code)


def test_self_launch_shared_device_smoke_flag(bench, monkeypatch):
    import subprocess
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd) or 0)
    monkeypatch.setenv("PFK_BENCH_SHARED_DEVICE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit):
        bench.self_launch(_args(2))
    assert "--nproc-per-node=2" in seen["cmd"]
