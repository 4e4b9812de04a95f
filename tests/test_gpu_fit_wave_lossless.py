"""k_fit_models_wave under a LOSSLESS bound decides by equality (mdb_fit.hip: 64 start points per round, eight values
per lane; PMC-Mean stands on eight equal values, Swing on eight values on the line through the first two; a division
only where the third point is within the reference's rounding of that line). What that path must get right, against
the CPU oracle byte for byte, with the wave kernel taking every chunk (MDB_FIT_WAVE=1), as the library chooses, and -
calls of a few chunks - in pieces through the small driver:
  * zeros of both signs in one run (PMC-Mean's value is +0.0: the sum starts there), runs of NaN and of either infinity
    (both fitters accept copies of a non-finite first value, swing.rs:113-125: the long way), runs shorter than 8;
  * points exactly on a line and one ulp off it, with timestamps from 0 and of epoch size (where the reference's own
    f64 arithmetic is coarse: the margin within which the exact test runs is wide), slopes of either sign;
  * timestamps that are loaded (irregular, exact in f64);
  * models that end with their chunk, chunks of 1..72 points."""

import numpy as np
import pytest

import cases
import oracle_lib as ora
from test_gpu_fit import assert_same_segments

pytestmark = pytest.mark.gpu

SWITCHES = ("MDB_FIT_WAVE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_SMALL", "MDB_FIT_GAP_MIN_VALUES",
            "MDB_FIT_WAVE_WINDOW_POINTS", "MDB_FIT_WAVE_POINTS_PER_STEP")
LOSSLESS = cases.error_bounds()["lossless"]


@pytest.fixture(autouse=True, params=["wave-per-chunk", "library", "small-driver"])
def mode(request, monkeypatch):
    for name in SWITCHES:
        monkeypatch.delenv(name, raising=False)
    if request.param == "wave-per-chunk":
        monkeypatch.setenv("MDB_FIT_WAVE", "1")
    if request.param == "library":
        monkeypatch.setenv("MDB_FIT_SMALL", "0")
    return request.param


def runs(rng, length, kinds):
    """A series made of runs of 3..40 points, each of one of `kinds`."""
    out, made = [], 0
    while made < length:
        n = int(rng.integers(3, 41))
        made += n
        kind = kinds[int(rng.integers(len(kinds)))]
        if kind == "zeros":
            run = np.where(rng.random(n) < 0.5, np.float32(0.0), np.float32(-0.0))
        elif kind == "negative-zeros":
            run = np.full(n, -0.0, np.float32)
        elif kind == "constant":
            run = np.full(n, np.float32(rng.normal() * 10.0 ** int(rng.integers(-30, 30))), np.float32)
        elif kind == "nan":
            run = np.full(n, np.nan, np.float32)
        elif kind == "inf":
            run = np.full(n, np.inf if rng.random() < 0.5 else -np.inf, np.float32)
        elif kind == "noise":
            run = rng.uniform(100.0, 200.0, n).astype(np.float32)
        elif kind == "line":
            run = (np.float32(rng.integers(-1000, 1000)) + np.arange(n, dtype=np.float32) * np.float32(rng.integers(-9, 10))).astype(np.float32)
        elif kind == "line-one-ulp-off":
            run = (np.float32(rng.integers(1, 1000)) + np.arange(n, dtype=np.float32) * np.float32(rng.integers(1, 10))).astype(np.float32)
            at = int(rng.integers(2, n))
            run[at] = np.nextafter(run[at], np.float32(np.inf if rng.random() < 0.5 else -np.inf))
        elif kind == "fractions":  # (a line whose values are not exact in f32: points fall on and off it)
            run = (rng.random() + np.arange(n) * rng.random() * 0.1).astype(np.float32)
        else:
            raise AssertionError(kind)
        out.append(run.astype(np.float32))
    return np.concatenate(out)[:length]


def fit_and_compare(hip, timestamps, values, offsets):
    import os
    expected = ora.compress_chunks(timestamps, values, offsets, LOSSLESS)
    hip.profile_enable(True)
    hip.profile_reset()
    got = hip.compress_chunks(timestamps, values, offsets, LOSSLESS)
    kernels = {name for name, (calls, _) in hip.profile().items() if calls > 0}
    hip.profile_enable(False)
    if os.environ.get("MDB_FIT_WAVE") == "1":
        assert "k_fit_models_wave" in kernels, kernels  # (the path under test took the call)
    assert_same_segments(got, expected)
    return expected


KINDS = {
    "zeros-and-constants": ["zeros", "negative-zeros", "constant", "noise"],
    "non-finite": ["nan", "inf", "constant", "noise", "zeros"],
    "lines": ["line", "line-one-ulp-off", "noise", "constant"],
    "fractions": ["fractions", "line", "noise"],
    "everything": ["zeros", "negative-zeros", "constant", "nan", "inf", "noise", "line", "line-one-ulp-off", "fractions"],
}


@pytest.mark.parametrize("first_time", [0, 1_658_671_178_037_000])
@pytest.mark.parametrize("kinds", sorted(KINDS))
def test_runs_of_every_kind(hip, kinds, first_time):
    rng = np.random.default_rng(len(kinds) + (1 if first_time else 0))
    lengths = [5000, 777, 64, 65, 8, 7, 1, 9000]
    values = np.concatenate([runs(rng, n, KINDS[kinds]) for n in lengths])
    timestamps = np.concatenate([first_time + np.arange(n, dtype=np.int64) * 1000 for n in lengths])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    expected = fit_and_compare(hip, timestamps, values, offsets)
    assert len(set(expected.model_type_id.tolist())) >= 2  # (models and MacaqueV segments both)


@pytest.mark.parametrize("kinds", ["lines", "everything"])
def test_loaded_timestamps(hip, kinds):
    """Irregular timestamps (every chunk's are loaded; all exact in f64): Swing's lines go through the real times."""
    rng = np.random.default_rng(7)
    lengths = [3000, 500, 71, 72, 16]
    values = np.concatenate([runs(rng, n, KINDS[kinds]) for n in lengths])
    timestamps = np.concatenate([1_700_000_000_000_000 + np.cumsum(rng.integers(1, 2000, n)).astype(np.int64) for n in lengths])
    # (a stretch of equal intervals in the middle of irregular ones: lines in time as well as in index)
    timestamps[1000:1400] = timestamps[1000] + np.arange(400, dtype=np.int64) * 250
    timestamps[1400:3000] = timestamps[1399] + np.cumsum(rng.integers(1, 2000, 1600)).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    fit_and_compare(hip, timestamps, values, offsets)


@pytest.mark.parametrize("length", list(range(1, 20)) + [63, 64, 65, 71, 72, 73, 127, 128, 129])
def test_models_that_end_with_their_chunk(hip, length):
    """One constant run, one line and one stretch of noise per call, each a chunk of `length` points: a model is
    accepted from 8 points on (compression.rs:238), the points of a shorter chunk are residuals of nothing - one
    MacaqueV segment."""
    rng = np.random.default_rng(length)
    chunks = [np.full(length, 42.5, np.float32), (7.0 + 3.0 * np.arange(length)).astype(np.float32),
              rng.uniform(1.0, 2.0, length).astype(np.float32),
              np.concatenate([rng.uniform(1.0, 2.0, length // 2).astype(np.float32), np.full(length - length // 2, -3.25, np.float32)])]
    values = np.concatenate(chunks)
    timestamps = np.concatenate([np.arange(length, dtype=np.int64) * 100 for _ in chunks])
    offsets = (np.arange(len(chunks) + 1) * length).astype(np.uint64)
    expected = fit_and_compare(hip, timestamps, values, offsets)
    if length >= 8:
        assert expected.model_type_id[0] == 0 and expected.model_type_id[1] == 1  # PMC-Mean, Swing


def test_many_chunks_of_the_acceptance_recipe(hip, mode):
    """compression.rs:733-863's recipe, 300 chunks of 4 000 points (no noise: constant and linear runs stand)."""
    if mode == "small-driver":
        pytest.skip("a call for the general driver")
    import datagen
    n_chunks, length = 300, 4000
    values = np.concatenate([datagen.mixed_series(length, 50 + c, None)[1] for c in range(n_chunks)])
    timestamps = np.tile(np.arange(length, dtype=np.int64) * 100, n_chunks)
    offsets = (np.arange(n_chunks + 1) * length).astype(np.uint64)
    expected = fit_and_compare(hip, timestamps, values, offsets)
    assert set(expected.model_type_id.tolist()) == {0, 1, 2}


@pytest.mark.parametrize("kinds", ["lines", "zeros-and-constants"])
def test_one_timestamp_for_every_point(hip, kinds):
    """Chunks whose points all carry the same timestamp (equally spaced, by an interval of 0): the line through two such
    points is not a number, and the reference's comparisons reject nothing against it (swing.rs:161-166) - whatever
    that makes of the chunk, the same here."""
    rng = np.random.default_rng(3)
    lengths = [300, 40, 9, 8]
    values = np.concatenate([runs(rng, n, KINDS[kinds]) for n in lengths])
    timestamps = np.concatenate([np.full(n, 1_000_000 * (k + 1), dtype=np.int64) for k, n in enumerate(lengths)])
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    fit_and_compare(hip, timestamps, values, offsets)


def test_more_chunks_than_the_lossy_bounds_hand_the_wave_kernel(hip, mode, monkeypatch):
    """Under a lossless bound the wave kernel takes a call of ANY number of chunks (it decides by equality and is the
    fastest fitter there; under lossy bounds it stops at 96 chunks per compute unit): 30 000 chunks of 150 to 200 points,
    the library's choice - the wave kernel - against the lane-per-chunk fitter on the same call, and the oracle on a sample."""
    if mode != "library":
        pytest.skip("the library's own choice is what is tested")
    rng = np.random.default_rng(99)
    n_chunks = 30_000
    lengths = rng.integers(150, 201, n_chunks)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    values = np.resize(runs(rng, 200_000, KINDS["everything"]), int(offsets[-1]))  # (the same 200 000 points over and over)
    timestamps = (np.arange(int(offsets[-1]), dtype=np.int64) - np.repeat(offsets[:-1].astype(np.int64), lengths)) * 1000
    hip.profile_enable(True)
    hip.profile_reset()
    chosen = hip.compress_chunks(timestamps, values, offsets, LOSSLESS)
    kernels = {name for name, (calls, _) in hip.profile().items() if calls > 0}
    hip.profile_enable(False)
    assert "k_fit_models_wave" in kernels, kernels
    monkeypatch.setenv("MDB_FIT_WAVE", "0")
    assert chosen.identical(hip.compress_chunks(timestamps, values, offsets, LOSSLESS))
    sample = 400
    end = int(offsets[sample])
    assert_same_segments(hip.compress_chunks(timestamps[:end], values[:end], offsets[:sample + 1], LOSSLESS),
                         ora.compress_chunks(timestamps[:end], values[:end], offsets[:sample + 1], LOSSLESS))
