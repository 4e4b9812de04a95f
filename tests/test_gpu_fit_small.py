"""The fit of a handful of chunks (mdb_fit.hip: fit_few_chunks - one upload, launches over upper bounds that read their
counts from device memory, one wave per PIECE of a chunk, one download) against the CPU oracle, byte for byte, and
against the general driver (MDB_FIT_SMALL=0), through mdb_compress_chunk_list / mdb_compress_chunks /
mdb_compress_series. The server's call shape: one finished buffer of 65 536 points
(crates/modelardb_server/src/storage/uncompressed_data_manager.rs:530-596)."""

import numpy as np
import pytest

import cases
import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb
from test_gpu_fit import assert_same_segments

pytestmark = pytest.mark.gpu

FIT_SWITCHES = ("MDB_FIT_WAVE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_GAP_MIN_VALUES",
                "MDB_FIT_DATA_BUFFER_BYTES", "MDB_FIT_DEBUG", "MDB_FIT_SMALL")


@pytest.fixture(autouse=True)
def no_switches(monkeypatch):
    for name in FIT_SWITCHES:
        monkeypatch.delenv(name, raising=False)


def oracle_of(chunks, eb):
    offsets = np.concatenate([[0], np.cumsum([len(v) for _, v in chunks])]).astype(np.uint64)
    timestamps = np.concatenate([t for t, _ in chunks]) if chunks else np.zeros(0, dtype=np.int64)
    values = np.concatenate([v for _, v in chunks]) if chunks else np.zeros(0, dtype=np.float32)
    return ora.compress_chunks(timestamps, values, offsets, eb)


def series(kind, length, seed):
    timestamps = 1_700_000_000_000 + np.arange(length, dtype=np.int64) * 250
    if kind == "sine":
        return timestamps, datagen.bench_series(seed, length)
    if kind == "mixed":
        return timestamps, datagen.mixed_series(length, 100 + seed, (1.0, 1.05) if seed % 2 else None)[1][:length]
    if kind == "noise":
        return timestamps, np.random.default_rng(seed).uniform(100.0, 200.0, length).astype(np.float32)
    if kind == "constant":
        return timestamps, np.full(length, 37.5, dtype=np.float32)
    raise AssertionError(kind)


@pytest.mark.parametrize("eb_name", ["lossless", "abs5", "rel5", "rel1", "abs0.01"])
@pytest.mark.parametrize("kind", ["sine", "mixed", "noise", "constant"])
def test_one_finished_buffer(hip, kind, eb_name):
    eb = cases.error_bounds()[eb_name]
    timestamps, values = series(kind, 65_536, 3)
    expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
    assert_same_segments(hip.compress_chunk_list([(timestamps, values)], eb), expected)
    assert_same_segments(hip.try_compress_univariate_time_series(timestamps, values, eb), expected)


@pytest.mark.parametrize("eb_name", ["lossless", "rel1", "abs5"])
def test_chunks_of_every_length(hip, eb_name, monkeypatch):
    """Chunks shorter than a model, than a wave's block, than a piece; around the piece size; empty ones in between."""
    eb = cases.error_bounds()[eb_name]
    lengths = [0, 1, 2, 3, 7, 8, 9, 63, 64, 65, 0, 255, 256, 257, 2047, 2048, 2049, 4095, 4097, 5000, 1, 12_345]
    chunks = [series(("sine", "mixed", "noise", "constant")[k % 4], n, k) for k, n in enumerate(lengths)]
    expected = oracle_of(chunks, eb)
    got = hip.compress_chunk_list(chunks, eb)
    assert_same_segments(got, expected)
    monkeypatch.setenv("MDB_FIT_SMALL", "0")
    assert_same_segments(hip.compress_chunk_list(chunks, eb), expected)


@pytest.mark.parametrize("n_chunks", [1, 4, 16, 64])
def test_several_finished_buffers(hip, n_chunks):
    eb = cases.error_bounds()["rel1"]
    chunks = [series(("sine", "mixed")[k % 2], 65_536, k) for k in range(n_chunks)]
    expected = oracle_of(chunks, eb)
    assert_same_segments(hip.compress_chunk_list(chunks, eb), expected)
    timestamps = np.concatenate([t for t, _ in chunks])
    values = np.concatenate([v for _, v in chunks])
    offsets = np.arange(0, (n_chunks + 1) * 65_536, 65_536, dtype=np.uint64)
    assert_same_segments(hip.compress_chunks(timestamps, values, offsets, eb), expected)


def test_what_the_path_leaves_to_the_general_driver(hip):
    """Timestamps that are not equally spaced, timestamps beyond 2^52, more chunks than the path takes: the same
    segments from the general driver."""
    eb = cases.error_bounds()["rel1"]
    timestamps, values = series("sine", 20_000, 5)
    irregular = timestamps.copy()
    irregular[1000:] += 13
    for ts in (irregular, timestamps + (1 << 53)):
        expected = ora.try_compress_univariate_time_series(ts, values, eb)
        assert_same_segments(hip.compress_chunk_list([(ts, values)], eb), expected)
    chunks = [series("mixed", 700, k) for k in range(65)]
    assert_same_segments(hip.compress_chunk_list(chunks, eb), oracle_of(chunks, eb))


def test_long_models_and_a_model_over_the_whole_buffer(hip):
    """A bound so generous that one model spans every piece (every piece's wave walks to the end of the buffer), and
    models of a few thousand points (chains that meet late)."""
    timestamps, values = series("sine", 65_536, 9)
    for eb in (mdb.error_bound("relative", 50.0), mdb.error_bound("relative", 10.0), mdb.error_bound("absolute", 3.0)):
        expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
        assert_same_segments(hip.compress_chunk_list([(timestamps, values)], eb), expected)


def test_non_finite_values_and_signed_zeros(hip):
    eb = cases.error_bounds()["rel5"]
    timestamps, values = series("mixed", 30_000, 2)
    values = values.copy()
    values[100:140] = np.nan
    values[5000] = np.inf
    values[5001:5010] = -np.inf
    values[9000:9100:2] = 0.0
    values[9001:9100:2] = -0.0
    for bound in (eb, cases.error_bounds()["lossless"]):
        expected = ora.try_compress_univariate_time_series(timestamps, values, bound)
        assert_same_segments(hip.compress_chunk_list([(timestamps, values)], bound), expected)


@pytest.mark.parametrize("eb_name", ["rel1", "lossless"])
def test_timestamps_around_two_to_the_52(hip, eb_name, monkeypatch):
    """A chunk that begins below 2^52 and ends above it, one that ends exactly on it, and negative mirrors: the general
    driver calls a chunk with ANY timestamp beyond +-2^52 "beyond" (k_fit_regular) and gives it the careful fitter, so
    the few-chunks path must leave exactly those chunks alone; either way the segments are the oracle's and the two
    drivers' are the same."""
    eb = cases.error_bounds()[eb_name]
    length, stride = 20_000, 250
    _, values = series("mixed", length, 4)
    limit = 1 << 52
    firsts = (limit - length * stride // 2, limit - (length - 1) * stride, limit - (length - 1) * stride + 1,
              -(limit - length * stride // 2), -limit, -limit - 1)
    for first in firsts:
        timestamps = first + np.arange(length, dtype=np.int64) * stride
        expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
        monkeypatch.delenv("MDB_FIT_SMALL", raising=False)
        small = hip.compress_chunk_list([(timestamps, values)], eb)
        monkeypatch.setenv("MDB_FIT_SMALL", "0")
        general = hip.compress_chunk_list([(timestamps, values)], eb)
        assert_same_segments(small, expected)
        assert_same_segments(general, expected)
