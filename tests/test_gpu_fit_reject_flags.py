"""Split mode under a lossy bound: the start points at which no model can begin, found for all points at once
(k_fit_reject_flags, modelardb-rs_amd/csrc/mdb_fit.hip) and passed over by the fitter without a point being fed, and the
probe that sends a call whose models are short in many places to split mode without one wave per chunk having tried it
(k_fit_models_wave's probe). Neither may change a byte: every case is the oracle's segments (compression.rs:181-262 with
pmc_mean.rs:58-76 and swing.rs:101-198 restated in oracle/), with the bits looked at, not looked at, and not made."""

import numpy as np
import pytest

import cases
import oracle_lib as ora
import modelardb_rs_amd as mdb
from test_gpu_fit import assert_same_segments

pytestmark = pytest.mark.gpu

BOUNDS = {
    "rel1": mdb.error_bound("relative", 1.0),
    "rel5": mdb.error_bound("relative", 5.0),
    "rel0.001": mdb.error_bound("relative", 0.001),
    "rel100": mdb.error_bound("relative", 100.0),
    "abs5": mdb.error_bound("absolute", 5.0),
    "abs0.01": mdb.error_bound("absolute", 0.01),
}


@pytest.fixture(autouse=True)
def _clean(monkeypatch):
    for name in ("MDB_FIT_PIECE_POINTS", "MDB_FIT_REJECT_FLAGS", "MDB_FIT_WAVE_PROBE", "MDB_FIT_WAVE", "MDB_FIT_LEAN"):
        monkeypatch.delenv(name, raising=False)


def _recipe(rng, n):
    """Runs of every texture next to each other: constants, lines, noise small and large against the bounds, values
    around zero, magnitudes at either end of f32, and a few values that are not numbers."""
    parts, made = [], 0
    while made < n:
        length = int(rng.integers(3, 400))
        kind = int(rng.integers(0, 9))
        level = float(rng.choice([0.0, 1e-30, 1e-3, 1.0, 100.0, -100.0, 1e6, 1e30]))
        if kind == 0:
            run = np.full(length, level)
        elif kind == 1:
            run = level + np.arange(length) * float(rng.uniform(-2, 2)) * max(abs(level), 1.0) * 1e-3
        elif kind == 2:
            run = level + rng.uniform(-1, 1, length) * max(abs(level), 1e-30) * 0.004   # inside a bound of 1 %
        elif kind == 3:
            run = level + rng.uniform(-1, 1, length) * max(abs(level), 1e-30) * 0.02    # about the bound
        elif kind == 4:
            run = level + rng.uniform(-1, 1, length) * max(abs(level), 1e-30) * 0.5     # far outside it
        elif kind == 5:
            run = rng.uniform(-10, 10, length)                                             # through zero
        elif kind == 6:
            run = np.where(rng.uniform(size=length) < 0.5, level, level * 1.011)          # two values a bound apart
        elif kind == 7:
            run = np.cumsum(rng.uniform(-1, 1, length)) + 100.0                           # a walk
        else:
            run = rng.uniform(-1e3, 1e3, length)
            run[rng.integers(0, length, 2)] = rng.choice([np.nan, np.inf, -np.inf])
        parts.append(run)
        made += length
    with np.errstate(over="ignore"):
        return np.concatenate(parts)[:n].astype(np.float32)


def _kernels(hip):
    return {name for name, (calls, _) in hip.profile().items() if calls > 0}


@pytest.mark.parametrize("start, interval", [(0, 100), (1_700_000_000_000_000, 1000), (-4_000_000_000_000_000, 7)])
@pytest.mark.parametrize("pieces", ["64", "128", "1024", "100"])
@pytest.mark.parametrize("bound", list(BOUNDS))
def test_passing_over_flagged_start_points_changes_no_byte(hip, monkeypatch, bound, pieces, start, interval):
    eb = BOUNDS[bound]
    rng = np.random.default_rng(abs(hash((bound, pieces, start))) % (1 << 32))
    lengths = [1, 7, 8, 9, 15, 16, 63, 64, 65, 127, 1000, 4097, 12_345]
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    values = _recipe(rng, int(offsets[-1]))
    timestamps = np.concatenate([start + interval * (7 * k + np.arange(length, dtype=np.int64)) for k, length in enumerate(lengths)])
    expected = ora.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_PIECE_POINTS", pieces)
    outcomes = {}
    for flags in ("1", None, "0"):   # looked at whatever their number, looked at if they are many, not made
        if flags is None:
            monkeypatch.delenv("MDB_FIT_REJECT_FLAGS", raising=False)
        else:
            monkeypatch.setenv("MDB_FIT_REJECT_FLAGS", flags)
        hip.profile_enable(True)
        hip.profile_reset()
        got = hip.compress_chunks(timestamps, values, offsets, eb)
        outcomes[flags] = _kernels(hip)
        hip.profile_enable(False)
        assert_same_segments(got, expected)
    assert "k_fit_models_split" in outcomes["1"]
    # (the bits are words of 64 points of a piece: pieces of 100 points have none)
    assert ("k_fit_reject_flags" in outcomes["1"]) == (int(pieces) % 64 == 0)
    assert "k_fit_reject_flags" not in outcomes["0"]


@pytest.mark.parametrize("bound", ["rel1", "abs5"])
def test_the_probe_sends_a_call_of_short_models_to_split_mode_untried(hip, monkeypatch, bound):
    eb = BOUNDS[bound]
    rng = np.random.default_rng(5)
    n_chunks, points = 1200, 4096
    offsets = np.arange(0, n_chunks * points + 1, points, dtype=np.uint64)
    timestamps = 1_000_000 + 50 * np.arange(n_chunks * points, dtype=np.int64)
    smooth = (300.0 + 100.0 * np.sin(np.arange(n_chunks * points) / 3000.0)).astype(np.float32)
    textures = {
        # noise of the bound's size in most windows - models of a dozen points, each a step of a wave's 64 lanes - and far
        # outside it in some: the probe's waves find no pace worth a wave per chunk
        "noisy": (smooth + np.where((np.arange(n_chunks * points) // 700) % 4 == 0, 12.0, 1.0)
                  * rng.uniform(-1, 1, n_chunks * points) * (4.0 if bound == "rel1" else 5.0)).astype(np.float32),
        # long models everywhere: every chunk gets its wave
        "smooth": smooth,
    }
    for texture, values in textures.items():
        expected = ora.compress_chunks(timestamps, values, offsets, eb)
        for probe in (None, "0"):
            if probe is None:
                monkeypatch.delenv("MDB_FIT_WAVE_PROBE", raising=False)
            else:
                monkeypatch.setenv("MDB_FIT_WAVE_PROBE", probe)
            hip.profile_enable(True)
            hip.profile_reset()
            got = hip.compress_chunks(timestamps, values, offsets, eb)
            kernels = _kernels(hip)
            hip.profile_enable(False)
            assert_same_segments(got, expected)
            assert ("k_fit_models_wave_probe" in kernels) == (probe is None), (texture, kernels)
            if probe is None and texture == "noisy":
                assert "k_fit_models_wave" not in kernels and {"k_fit_models_split", "k_fit_reject_flags", "k_fit_walk"} <= kernels, kernels
            if texture == "smooth":
                assert "k_fit_models_wave" in kernels and "k_fit_models_split" not in kernels, kernels


@pytest.mark.parametrize("texture", ["noisy", "smooth"])
def test_the_probe_with_timestamps_that_are_loaded(hip, texture):
    """Irregular timestamps (exact as f64: the wave kernel and the lean fitter take them loaded): the probe's waves read
    them like any other, the call goes one way, the flags stay out of it (they are the values-only fitter's) - and the
    segments are the oracle's."""
    eb = BOUNDS["rel1"]
    rng = np.random.default_rng(11)
    n_chunks, points = 600, 4096
    offsets = np.arange(0, n_chunks * points + 1, points, dtype=np.uint64)
    timestamps = np.cumsum(rng.integers(1, 400, n_chunks * points)).astype(np.int64) + 1_600_000_000_000
    values = 300.0 + 100.0 * np.sin(np.arange(n_chunks * points) / 3000.0)
    if texture == "noisy":
        values = values + rng.uniform(-4.0, 4.0, n_chunks * points)
    values = values.astype(np.float32)
    expected = ora.compress_chunks(timestamps, values, offsets, eb)
    hip.profile_enable(True)
    hip.profile_reset()
    got = hip.compress_chunks(timestamps, values, offsets, eb)
    kernels = _kernels(hip)
    hip.profile_enable(False)
    assert_same_segments(got, expected)
    assert "k_fit_models_wave_probe" in kernels and "k_fit_reject_flags" not in kernels, kernels
    assert ("k_fit_models_split" in kernels) == (texture == "noisy"), kernels
    assert ("k_fit_models_wave" in kernels) == (texture == "smooth"), kernels
