"""bench.py's launch contract, checked without a GPU: `--gpus N` outside a launcher starts N ranks as a
CHILD `torch.distributed.run` before anything touches the GPU, relays the child's line and leaves with its
code; inside a launcher (WORLD_SIZE set) it does not launch again; N = 1 launches nothing."""

import os
import subprocess
import sys
import types

import pytest

import bench


class _Exit(Exception):
    pass


def _args(gpus):
    return types.SimpleNamespace(gpus=gpus)


def test_gpus_n_launches_n_ranks_as_a_child_process(monkeypatch, capsys):
    calls = []

    def fake_run(command, env=None, stdout=None, text=None):
        calls.append((command, env))
        return types.SimpleNamespace(stdout='{"metric": "gridded values/sec", "n_gpus": 4}\n', returncode=3)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as left:
        bench.launch_ranks_if_needed(_args(4))
    assert left.value.code == 3                                   # the child's code
    assert capsys.readouterr().out == '{"metric": "gridded values/sec", "n_gpus": 4}\n'
    (command, env), = calls
    assert command[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in command and "--nnodes=1" in command
    assert command[command.index("--master-addr") + 1] == "127.0.0.1"
    assert command[-4:] == ["--gpus", "4", "--steps", "2"] and command[-5] == os.path.abspath(bench.__file__)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("environment", [{"WORLD_SIZE": "4"}, {"RANK": "0"}])
def test_no_second_launch_inside_a_launcher(monkeypatch, environment):
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("launched")))
    for name, value in environment.items():
        monkeypatch.setenv(name, value)
    assert bench.launch_ranks_if_needed(_args(4)) is None


def test_one_gpu_launches_nothing(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("launched")))
    assert bench.launch_ranks_if_needed(_args(1)) is None


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.init_distributed(_args(8))
