"""k_fit_models_lean with its groups of 64 chunks ROTATING over the waves (mdb_fit.hip: LeanRotation, rotation_take /
rotation_give, lean_saved_store / lean_saved_load): a wave steps a group through a stretch of steps, writes the 64
lanes' fitters to memory, queues the group and takes the one that has waited longest. What comes out must not depend
on where the stretches end - against the CPU oracle byte for byte, and against the plain kernel on the same call, with
stretches of one step (every fitter through memory between any two points), of a few, and of the library's own
length; with calls of two groups (one wave alternates between them) up to some hundred; with chunks that are empty,
shorter than a stretch, ragged; under each kind of error bound (compression.rs:89-218 is the loop being cut)."""

import numpy as np
import pytest

import cases
import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb
from test_gpu_fit import assert_same_segments

pytestmark = pytest.mark.gpu

SWITCHES = ("MDB_FIT_WAVE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_SMALL", "MDB_FIT_ROTATE",
            "MDB_FIT_ROTATE_STEPS", "MDB_FIT_ROTATE_MAX_NAPS", "MDB_FIT_GAP_MIN_VALUES")


@pytest.fixture(autouse=True)
def lane_per_chunk(monkeypatch):
    for name in SWITCHES:
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setenv("MDB_FIT_WAVE", "0")          # (not a wave per chunk,
    monkeypatch.setenv("MDB_FIT_PIECE_POINTS", "1")  # not pieces,
    monkeypatch.setenv("MDB_FIT_SMALL", "0")         # not the few-chunks driver: k_fit_models_lean)


def call_of(lengths, seed):
    """Chunks of the bench's mixture of models (datagen.bench_series) with regular timestamps."""
    values = [datagen.bench_series(seed + k, n, 0x4D44425F52454631) if n else np.zeros(0, np.float32) for k, n in enumerate(lengths)]
    timestamps = np.concatenate([np.arange(n, dtype=np.int64) * 1000 for n in lengths]) if lengths else np.zeros(0, np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    return timestamps, np.concatenate(values), offsets


def kernels_run(hip):
    return {name for name, (calls, _) in hip.profile().items() if calls > 0}


@pytest.mark.parametrize("steps", ["1", "5", "37", None])
@pytest.mark.parametrize("eb_name", ["rel1", "abs0.01", "lossless"])
def test_rotating_groups_fit_as_the_oracle_does(hip, monkeypatch, steps, eb_name):
    eb = cases.error_bounds()[eb_name]
    rng = np.random.default_rng(12)
    lengths = [int(n) for n in rng.integers(1, 900, 300)]
    for at, n in ((0, 0), (17, 0), (63, 1), (64, 2), (130, 3000), (299, 0)):
        lengths[at] = n
    timestamps, values, offsets = call_of(lengths, 100)
    expected = ora.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_ROTATE", "1")
    if steps is not None:
        monkeypatch.setenv("MDB_FIT_ROTATE_STEPS", steps)
    hip.profile_enable(True)
    hip.profile_reset()
    got = hip.compress_chunks(timestamps, values, offsets, eb)
    assert "k_fit_models_lean" in kernels_run(hip)
    hip.profile_enable(False)
    assert_same_segments(got, expected)


@pytest.mark.parametrize("n_chunks", [65, 128, 129, 1000, 20_000])
def test_rotation_against_the_plain_kernel(hip, monkeypatch, n_chunks):
    """The same call with MDB_FIT_ROTATE=0 and =1 (stretches of 23 steps): identical columns, whatever the number of
    groups - two (one wave, the groups by turns), a few, and more than the device has SIMDs."""
    eb = cases.error_bounds()["rel1"]
    length = 700 if n_chunks <= 1000 else 150
    timestamps, values, offsets = call_of([length - (k % 7) for k in range(n_chunks)], 7)
    monkeypatch.setenv("MDB_FIT_ROTATE", "0")
    plain = hip.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_ROTATE", "1")
    monkeypatch.setenv("MDB_FIT_ROTATE_STEPS", "23")
    rotated = hip.compress_chunks(timestamps, values, offsets, eb)
    assert rotated.identical(plain)
    if n_chunks <= 1000:
        assert_same_segments(rotated, ora.compress_chunks(timestamps, values, offsets, eb))


def test_rotation_chosen_by_the_library(hip, monkeypatch):
    """No switch: a call whose groups of 64 chunks are more than the device's SIMDs, and not a whole number per SIMD,
    rotates by itself (2.3 groups per SIMD here, as in the bench's call) - and fits what the plain kernel fits."""
    info = hip.device_info()
    simds = 4 * info["compute_units"]
    n_chunks = 64 * (2 * simds + simds // 3)
    eb = cases.error_bounds()["rel1"]
    timestamps, values, offsets = call_of([64] * n_chunks, 3)
    auto = hip.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_ROTATE", "0")
    plain = hip.compress_chunks(timestamps, values, offsets, eb)
    assert auto.identical(plain)
    sample = 64 * 40
    expected = ora.compress_chunks(timestamps[:64 * sample], values[:64 * sample], offsets[:sample + 1], eb)
    assert len(expected) > 0
    got = hip.compress_chunks(timestamps[:64 * sample], values[:64 * sample], offsets[:sample + 1], eb)
    assert_same_segments(got, expected)


def test_the_bench_shape_with_stretches_of_one_step(hip, monkeypatch):
    """The headline's call in groups - 2 391 of them (153 000 chunks), which is what rotates by itself on 1 024 SIMDs -
    with every fitter through memory between any two points (stretches of ONE step: some 150 handovers per group and
    2 391 groups in the queue at once): identical to the plain kernel's columns, and to the oracle's on a sample."""
    n_chunks = 153_000
    eb = cases.error_bounds()["rel1"]
    timestamps, values, offsets = call_of([150 - (k % 5) for k in range(n_chunks)], 11)
    monkeypatch.setenv("MDB_FIT_ROTATE", "0")
    plain = hip.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_ROTATE", "1")
    monkeypatch.setenv("MDB_FIT_ROTATE_STEPS", "1")
    rotated = hip.compress_chunks(timestamps, values, offsets, eb)
    assert rotated.identical(plain)
    sample = 500
    end = int(offsets[sample])
    assert_same_segments(hip.compress_chunks(timestamps[:end], values[:end], offsets[:sample + 1], eb),
                         ora.compress_chunks(timestamps[:end], values[:end], offsets[:sample + 1], eb))


def test_a_wave_that_waits_too_long_fails_the_call(hip, monkeypatch):
    """rotation_take's wait is bounded: with a bound of ONE nap (some 14 us; a minute by default) the waves that wait
    at the queue's end for the last groups give up, and the call fails with the library's error text instead of
    hanging - and the next call on the same context works."""
    eb = cases.error_bounds()["rel1"]
    timestamps, values, offsets = call_of([60_000] * 200, 5)  # 4 groups, 3 waves: the last stretches leave waves waiting
    monkeypatch.setenv("MDB_FIT_ROTATE", "1")
    monkeypatch.setenv("MDB_FIT_ROTATE_STEPS", "50000")
    monkeypatch.setenv("MDB_FIT_ROTATE_MAX_NAPS", "1")
    with pytest.raises(mdb.HipError, match="rotating fit waited"):
        hip.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.delenv("MDB_FIT_ROTATE_MAX_NAPS")
    again = hip.compress_chunks(timestamps, values, offsets, eb)
    monkeypatch.setenv("MDB_FIT_ROTATE", "0")
    assert again.identical(hip.compress_chunks(timestamps, values, offsets, eb))
