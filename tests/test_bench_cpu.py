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


def _no_constants(name):
    raise AssertionError(f"{name} on the bench line: not strict JSON")


def test_the_line_of_a_full_run_is_compact_and_strict_json():
    """compact_line() over everything a default run measures (round 5's detail, 21.9 KB): the contract's keys, `config`,
    `roofline`, `cpu_baseline` and the scalars of the secondary blocks in under 4 KB, no NaN / Infinity."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("bench_default.json", "bench_timed_fit.json", "bench_config5_shape_1gpu.json"):
        detail = json.load(open(os.path.join(root, "tests", "golden", "bench_detail_r05", name)))
        detail["config"]["libraries"] = {"libamdhip64": ["/opt/rocm-7.2.0/lib/libamdhip64.so.7.2.70200",
                                                         "/usr/local/lib/python3.10/dist-packages/torch/lib/libamdhip64.so"],
                                         "librccl": ["/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so"]}
        detail["aggregates_nan_probe"] = float("nan")
        text = bench.compact_line(detail)
        assert len(text.encode()) <= bench.COMPACT_LIMIT_BYTES <= 4096 and "\n" not in text
        line = json.loads(text, parse_constant=_no_constants)
        for key in bench.CONTRACT_KEYS:
            assert line[key] == detail[key], key
        assert line["config"]["workload"] == detail["config"]["workload"]
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "kernel_ms"}
        assert abs(line["roofline"]["frac"] - detail["roofline"]["frac"]) < 1e-5
        assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        if name == "bench_default.json":
            assert line["config"]["segment_mix"] == detail["config"]["segment_mix"]
            assert line["roofline"]["traffic"] is not None
            assert {"fit_points_per_s", "fit_kernel_ms", "fit_frac_of_hbm", "mixed_grid_frac_lossless", "mixed_grid_frac_1pct",
                    "mixed_aggregates_frac_lossless", "range_aggregates_frac_of_hbm"} <= set(line["also"])


def test_an_overlong_line_is_refused():
    with pytest.raises(SystemExit, match="bytes"):
        bench.compact_line({"metric": "m", "config": {"workload": "w", "segment_mix": {f"k{i}": i for i in range(600)}}})


def _run_two_ranks(fail_on=None):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    environment = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if fail_on:
        environment["FAIL_ON"] = fail_on
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           os.path.join(root, "tests", "stub", "run_bench_orchestration.py")],
                          env=environment, capture_output=True, text=True, timeout=300)


def test_two_ranks_over_gloo_meet_close_in_order_and_print_one_line():
    """bench.orchestrate() at WORLD_SIZE = 2 (gloo, a canned workload): warm-up, barrier, K timed steps, barrier,
    MAX over ranks; the report's collective on every rank, rank 0's tail alone while rank 1 waits at the meeting
    point; every rank closes its workload; ONE line, n_gpus 2, both states seen, exit code 0."""
    import json
    done = _run_two_ranks()
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, done.stdout
    assert done.stdout.strip().splitlines()[-1] == lines[0]      # the LAST line of stdout is the line
    assert len(lines[0].encode()) <= 6000                         # (round 5's 21.9 KB line was not read back)
    line = json.loads(lines[0], parse_constant=_no_constants)
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and line["steps"] == 3
    assert line["also"]["aggregates_count"] == 2_000_000
    # everything that was measured is in the detail file the line names
    detail = json.load(open(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "tests", "stub", line["detail"])))
    assert detail["aggregates"]["result"]["count"] == 2_000_000 and detail["value"] == line["value"]
    assert line["value"] > 0 and line["ms_per_step_per_rank"]["max"] >= line["ms_per_step_per_rank"]["min"]
    assert "rank 0 closed" in done.stderr and "rank 1 closed" in done.stderr


@pytest.mark.parametrize("fail_on", ["1:step", "0:tail"])
def test_a_failing_rank_ends_the_job_non_zero_without_a_line(fail_on):
    """Whichever rank fails, wherever: no JSON line, a non-zero exit, nobody hangs (the survivors leave through
    the meeting point, or through the process group's timeout when the failing rank never gets there)."""
    done = _run_two_ranks(fail_on)
    assert done.returncode != 0
    assert not [line for line in done.stdout.splitlines() if line.startswith("{")]
    assert "asked to" in done.stderr
