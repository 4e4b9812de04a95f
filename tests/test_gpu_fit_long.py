"""Long MacaqueV-only segments cut into blocks with a wave each (mdb_fit.hip: k_fit_long*, k_fit_long_stitch): every
block is sized from an empty window / nothing stored, one wave per segment then walks the blocks with the true state,
and the blocks are written from it - against the CPU oracle byte for byte, through both drivers (the few-chunks path and
the general one), with the default block size on long streams and with blocks of 64 and 128 values on short ones so
that every kind of block boundary occurs: inside runs of repeated values, inside runs that keep their window, at NaNs,
with lossy bounds whose stored-value chain has to fall in with the true one. BASELINE's configuration 1 (one series of
10^6 points under a lossless bound) is the shape this is for (macaque_v.rs:76-214)."""

import numpy as np
import pytest

import cases
import oracle_lib as ora
import modelardb_rs_amd as mdb
from test_gpu_fit import assert_same_segments

pytestmark = pytest.mark.gpu

SWITCHES = ("MDB_FIT_WAVE", "MDB_FIT_PIECE_POINTS", "MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_GAP_MIN_VALUES",
            "MDB_FIT_DATA_BUFFER_BYTES", "MDB_FIT_DEBUG", "MDB_FIT_SMALL", "MDB_FIT_GAP_LONG_MIN_VALUES",
            "MDB_FIT_GAP_BLOCK_VALUES", "MDB_FIT_GAP_LONG_BELOW_WAVES", "MDB_FIT_GAP_ONCE")


@pytest.fixture(autouse=True)
def no_switches(monkeypatch):
    for name in SWITCHES:
        monkeypatch.delenv(name, raising=False)


@pytest.fixture(params=["few-chunks", "general"])
def driver(request, monkeypatch):
    if request.param == "general":
        monkeypatch.setenv("MDB_FIT_SMALL", "0")
    return request.param


def timestamps_of(length):
    return 1_700_000_000_000 + np.arange(length, dtype=np.int64) * 1000


def stream(kind, length, seed):
    """Values no PMC-Mean or Swing model of eight points fits under the bounds used here."""
    rng = np.random.default_rng(seed)
    if kind == "noise":                     # every code opens a window or nearly
        return rng.uniform(100.0, 200.0, length).astype(np.float32)
    if kind == "coarse":                    # few distinct mantissas: long runs inside one window, many repeats
        return (rng.integers(0, 4, length) * 64.0 + rng.integers(0, 2, length) * 1024.0).astype(np.float32)
    if kind == "runs":
        # Runs of one to five values around a level (too short for a model), the levels far apart: within a run the
        # values differ in their last bits only (one window serves the run; under a lossy bound the run stores its
        # first value again and again), every third run repeats its value exactly.
        levels = 100.0 + 1000.0 * (np.cumsum(rng.integers(1, 6, length)) % 6)   # (never the same level twice in a row)
        lengths = rng.integers(1, 6, length)
        values = np.repeat(levels, lengths)[:length]
        jitter = rng.uniform(-0.002, 0.002, length) * np.repeat(np.arange(length) % 3 != 0, lengths)[:length]
        return (values + jitter).astype(np.float32)
    if kind == "specials":
        values = rng.uniform(-1e6, 1e6, length).astype(np.float32)
        values[rng.random(length) < 0.05] = np.nan
        values[rng.random(length) < 0.02] = np.inf
        values[rng.random(length) < 0.02] = -np.inf
        values[rng.random(length) < 0.05] = 0.0
        values[rng.random(length) < 0.05] = -0.0
        return values
    if kind == "quiet-then-noise":          # a stretch of equal values (no window is ever opened in its blocks)
        values = rng.uniform(1.0, 2.0, length).astype(np.float32)
        values[length // 5: 3 * length // 5: 2] = 5.0e6
        values[length // 5 + 1: 3 * length // 5: 2] = -5.0e6
        values[2 * length // 5: 2 * length // 5 + 700] = values[2 * length // 5]
        return values
    raise AssertionError(kind)


def check(hip, chunks, eb):
    offsets = np.concatenate([[0], np.cumsum([len(v) for _, v in chunks])]).astype(np.uint64)
    expected = ora.compress_chunks(np.concatenate([t for t, _ in chunks]), np.concatenate([v for _, v in chunks]), offsets, eb)
    got = hip.compress_chunk_list(chunks, eb)
    assert_same_segments(got, expected)
    return expected


@pytest.mark.parametrize("length", [8193, 65_536, 300_001])
def test_a_long_lossless_stream_of_noise(hip, driver, length):
    values = stream("noise", length, length)
    expected = check(hip, [(timestamps_of(length), values)], cases.LOSSLESS)
    assert len(expected) == 1 and expected.model_type_id[0] == mdb.MDB_MACAQUE_V_ID


@pytest.mark.parametrize("block", ["64", "128", "1024"])
@pytest.mark.parametrize("kind", ["noise", "coarse", "runs", "specials", "quiet-then-noise"])
def test_short_streams_in_many_blocks(hip, driver, kind, block, monkeypatch):
    monkeypatch.setenv("MDB_FIT_GAP_LONG_MIN_VALUES", "256")
    monkeypatch.setenv("MDB_FIT_GAP_BLOCK_VALUES", block)
    for length in (256, 257, 320, 321, 1000, 4097, 20_000):
        values = stream(kind, length, length + 7)
        check(hip, [(timestamps_of(length), values)], cases.LOSSLESS)


@pytest.mark.parametrize("eb_name", ["abs0.01", "rel1", "abs5", "rel5"])
@pytest.mark.parametrize("kind", ["noise", "coarse", "runs", "specials", "quiet-then-noise"])
def test_lossy_bounds_in_many_blocks(hip, driver, kind, eb_name, monkeypatch):
    """Under a lossy bound the value stored before a block is unknown to the block's wave as well: its chain of stored
    values falls in with the true one at the first value both store anew."""
    monkeypatch.setenv("MDB_FIT_GAP_LONG_MIN_VALUES", "256")
    monkeypatch.setenv("MDB_FIT_GAP_BLOCK_VALUES", "64")
    eb = cases.error_bounds()[eb_name]
    for length in (300, 2049, 30_000):
        values = stream(kind, length, length + 11)
        check(hip, [(timestamps_of(length), values)], eb)


def test_several_chunks_some_long_some_not(hip, driver):
    lengths = [50_000, 300, 9000, 0, 8192, 8191, 70_000, 12]
    chunks = [(timestamps_of(n), stream(("noise", "coarse", "runs", "specials")[k % 4], n, k)) for k, n in enumerate(lengths)]
    for eb in (cases.LOSSLESS, cases.error_bounds()["abs0.01"]):
        check(hip, chunks, eb)


def test_long_segments_between_models(hip, driver, monkeypatch):
    """Long stretches of noise between stretches a model fits: MacaqueV segments of their own in the middle of a chunk
    (compression.rs:329-349), sized and written by blocks while the others go their usual way."""
    monkeypatch.setenv("MDB_FIT_GAP_LONG_MIN_VALUES", "512")
    monkeypatch.setenv("MDB_FIT_GAP_BLOCK_VALUES", "128")
    rng = np.random.default_rng(5)
    parts = []
    for k in range(6):
        parts.append(np.full(400, 10.0 * k, dtype=np.float32))
        parts.append(rng.uniform(0.0, 1000.0, 700 + 300 * k).astype(np.float32))
    values = np.concatenate(parts)
    for eb in (cases.LOSSLESS, cases.error_bounds()["rel1"]):
        expected = check(hip, [(timestamps_of(len(values)), values)], eb)
        assert (expected.model_type_id == mdb.MDB_MACAQUE_V_ID).sum() >= 5


def test_irregular_timestamps_of_a_long_segment(hip, monkeypatch):
    monkeypatch.setenv("MDB_FIT_GAP_LONG_MIN_VALUES", "256")
    monkeypatch.setenv("MDB_FIT_GAP_BLOCK_VALUES", "64")
    length = 5000
    timestamps = timestamps_of(length).copy()
    timestamps[3333:] += 17            # (found by a block in the middle: the segment's timestamps are not equally spaced)
    check(hip, [(timestamps, stream("noise", length, 3))], cases.LOSSLESS)
    check(hip, [(timestamps_of(length), stream("noise", length, 3))], cases.LOSSLESS)


def test_the_switch_that_turns_the_blocks_off(hip, driver, monkeypatch):
    monkeypatch.setenv("MDB_FIT_GAP_LONG_MIN_VALUES", "off")
    check(hip, [(timestamps_of(20_000), stream("noise", 20_000, 1))], cases.LOSSLESS)


def test_enough_streams_for_a_wave_each(hip, monkeypatch):
    """A call with so many MacaqueV-only segments that a wave each occupies the device leaves the long ones whole (the
    general driver decides by the count; here the line is moved so that two streams are "enough")."""
    monkeypatch.setenv("MDB_FIT_SMALL", "0")
    monkeypatch.setenv("MDB_FIT_GAP_LONG_BELOW_WAVES", "2")
    chunks = [(timestamps_of(n), stream(kind, n, n)) for kind, n in (("noise", 20_000), ("coarse", 9000), ("runs", 300), ("noise", 8192))]
    for eb in (cases.LOSSLESS, cases.error_bounds()["abs0.01"]):
        check(hip, chunks, eb)


@pytest.mark.parametrize("once", ["0", "1"])
def test_macaque_v_segments_encoded_once_or_twice(hip, monkeypatch, once):
    """The general driver's waves encode a MacaqueV-only segment into a staging place and copy it once its place is
    known (MDB_FIT_GAP_ONCE=1; what the library does under a lossy bound), or size it and encode it again (=0; under a
    lossless bound): the same bytes either way, for streams of every length around the copy's word boundaries (the
    copy moves aligned words; 13 bytes is the shortest stream that does not live in its view) and next to short ones
    that one lane encodes."""
    monkeypatch.setenv("MDB_FIT_SMALL", "0")
    monkeypatch.setenv("MDB_FIT_GAP_ONCE", once)
    monkeypatch.setenv("MDB_FIT_GAP_MIN_VALUES", "2")
    lengths = list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 300, 1000, 4097, 0, 20_000]
    chunks = [(timestamps_of(n), stream(("noise", "coarse", "runs", "specials")[k % 4], n, 100 + k)) for k, n in enumerate(lengths)]
    for eb in (cases.LOSSLESS, cases.error_bounds()["abs0.01"], cases.error_bounds()["rel1"]):
        check(hip, chunks, eb)
