"""Parity of the HIP grid() path with the CPU oracle, through the C ABI (mdb_grid_batch).

Bar: timestamps bit-exact and values bit-exact against the oracle's reconstruction of the same
segments (the segments are lossy w.r.t. the raw data within epsilon; decoding them is exact)."""

import numpy as np
import pytest

import cases
import oracle_lib as ora
import modelardb_rs_amd as mdb

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[None, "1024", "8", "ts-one-lane", "ts-general", "ts-no-jumps", "host-cursors"],
                ids=["mv-default", "mv-from-1024-values", "mv-from-8-values", "timestamps-one-lane-per-segment",
                     "timestamps-general-kernel-only", "timestamps-no-jump-lists", "mv-cursors-by-host-threads"])
def macaque_decoder(request, monkeypatch):
    """Every grid test runs with the parallel MacaqueV decoder (mdb_macaque_parallel.hpp) at its
    default threshold, switched off (one lane per stream only) and forced onto every stream of at
    least 8 values; and with irregular timestamps decoded one lane per 256-bit piece of a stream
    (k_grid_timestamps, the default: its sparse flavour first where the batch has few points per piece, or
    the general one alone) and one lane per segment (k_grid_serial); and without the jump lists that let
    k_grid_tiles write the timestamps of a fixed rate with the odd gap (and with the streams counted in the
    order of their segments instead of by length); and with cursors into the MacaqueV streams left by the
    call's host threads (mv_host_index)."""
    monkeypatch.delenv("MDB_GRID_TS_PIECES", raising=False)
    monkeypatch.delenv("MDB_GRID_TS_SPARSE", raising=False)
    monkeypatch.delenv("MDB_GRID_TS_JUMPS", raising=False)
    monkeypatch.delenv("MDB_GRID_TS_SORT", raising=False)
    monkeypatch.delenv("MDB_GRID_MV_HOST_MIN_VALUES", raising=False)
    if request.param is None:
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
    elif request.param == "host-cursors":
        # A call over host batches has its host threads walk the MacaqueV streams of segments with regular timestamps
        # (by default the long ones, here all of them) and decodes those piece by piece from the cursors they leave;
        # the parallel decoder from 8 values on for what is left.
        monkeypatch.setenv("MDB_GRID_MV_HOST_MIN_VALUES", "1")
        monkeypatch.setenv("MDB_GRID_MV_MIN_VALUES", "8")
    elif request.param == "ts-one-lane":
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
        monkeypatch.setenv("MDB_GRID_TS_PIECES", "off")
    elif request.param == "ts-general":
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
        monkeypatch.setenv("MDB_GRID_TS_SPARSE", "0")
    elif request.param == "ts-no-jumps":
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
        monkeypatch.setenv("MDB_GRID_TS_JUMPS", "0")
        monkeypatch.setenv("MDB_GRID_TS_SORT", "0")
    else:
        monkeypatch.setenv("MDB_GRID_MV_MIN_VALUES", request.param)
    return request.param


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("eb_name", ["lossless", "abs5", "rel5", "rel1"])
def test_grid_matches_oracle_on_synthetic_series(hip, eb_name, irregular):
    eb = cases.error_bounds()[eb_name]
    timestamps, values, batch = cases.mixed_batch(eb, irregular, seed=11)
    expected = ora.grid_batch(batch)
    assert hip.grid_count(batch) == len(expected[0])
    got = hip.grid_batch(batch)
    cases.assert_grid_equal(got, expected)
    assert np.array_equal(got[2], expected[2])      # rows per segment
    assert got[3] == expected[3]                    # GridStreamMetrics counters
    assert np.array_equal(got[0], timestamps)       # round trip: timestamps bit-exact


def test_grid_edge_cases(hip):
    for eb_name in ("lossless", "rel5", "abs5"):
        batch = cases.edge_case_batch(cases.error_bounds()[eb_name])
        expected = ora.grid_batch(batch)
        got = hip.grid_batch(batch)
        cases.assert_grid_equal(got, expected)
        assert np.array_equal(got[2], expected[2])
        assert got[3] == expected[3]


def test_grid_each_edge_case_alone(hip):
    for name, ts, values in cases.edge_case_series():
        batch = ora.try_compress_univariate_time_series(ts, values, cases.LOSSLESS)
        got = hip.grid_batch(batch)
        assert np.array_equal(got[0], ts), name
        assert np.array_equal(got[1].view(np.uint32), values.view(np.uint32)), name


def test_grid_empty_batch(hip):
    batch = mdb.SegmentBatch.from_rows([])
    assert hip.grid_count(batch) == 0
    ts, values, rows, metrics = hip.grid_batch(batch)
    assert len(ts) == 0 and len(values) == 0 and len(rows) == 0
    assert metrics["rows_created"] == 0


def test_grid_reference_size_batches(hip):
    # DataFusion hands GridExec batches of 8192 segment rows (SessionConfig default).
    eb = cases.error_bounds()["rel1"]
    _, _, batch = cases.mixed_batch(eb, False, seed=12, length=400_000, noise=None)
    expected = ora.grid_batch(batch)
    for start in range(0, len(batch), 8192):
        part = batch.slice(start, min(len(batch), start + 8192))
        cases.assert_grid_equal(hip.grid_batch(part), ora.grid_batch(part))
    cases.assert_grid_equal(hip.grid_batch(batch), expected)


def test_grid_many_tiny_segments_in_one_tile(hip):
    # > 1024 segments inside one 4096-point output tile exercises the global-search path.
    rows = [(0, 10 * i, 10 * i, b"", float(i), float(i), b"", b"") for i in range(5000)]
    rows += [(0, 100_000 + 10 * i, 100_000 + 10 * i + 5, b"", 1.0, 1.0, b"", b"") for i in range(3000)]
    batch = mdb.SegmentBatch.from_rows(rows)
    cases.assert_grid_equal(hip.grid_batch(batch), ora.grid_batch(batch))


def test_grid_one_huge_segment_and_neighbours(hip):
    n = 1_000_003
    rows = [
        (0, 0, 90, bytes([10]), 2.5, 2.5, b"", b""),
        (1, 1000, 1000 + (n - 1) * 10, n.to_bytes(3, "big"), -5.0, 5.0, b"", b""),
        (0, 20_000_000, 20_000_070, bytes([8]), 7.5, 7.5, b"", b""),
    ]
    batch = mdb.SegmentBatch.from_rows(rows)
    cases.assert_grid_equal(hip.grid_batch(batch), ora.grid_batch(batch))


def test_grid_multi_buffer_binary_views(hip):
    # Arrow may spread out-of-line payloads over several variadic buffers.
    eb = cases.error_bounds()["lossless"]
    _, _, batch = cases.mixed_batch(eb, True, seed=13, length=5_000)
    expected = ora.grid_batch(batch)
    for column_name in ("timestamps", "values", "residuals"):
        column = getattr(batch, column_name)
        items = column.to_bytes_list()
        views = column.views.copy()
        buffers = [np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
        for i, item in enumerate(items):
            if len(item) <= 12:
                continue
            which = i % 2
            views[i, 8:12] = np.frombuffer(np.int32(which).tobytes(), dtype=np.uint8)
            views[i, 12:16] = np.frombuffer(np.int32(len(buffers[which])).tobytes(), dtype=np.uint8)
            buffers[which] = np.concatenate([buffers[which], np.frombuffer(item, dtype=np.uint8)])
        setattr(batch, column_name, mdb.BinaryViewColumn(views, buffers))
    cases.assert_grid_equal(hip.grid_batch(batch), expected)


@pytest.mark.parametrize("rows,message", [
    ([(3, 0, 10, b"", 0.0, 0.0, b"", b"")], "unknown model type"),
    ([(0, 0, 10, bytes([1]), 0.0, 0.0, b"", b"")], "timestamps"),             # length 1: div by zero
    ([(0, 0, 10, bytes([5]), 0.0, 0.0, b"ab", b"")], "values"),               # types.rs:316-318
    ([(1, 0, 10, bytes([5]), 0.0, 0.0, b"abc", b"")], "values"),              # types.rs:405
    ([(2, 0, 10, bytes([5]), 0.0, 0.0, b"", b"")], "values"),                 # macaque_v.rs:279-280
    ([(0, 0, 40, bytes([5]), 0.0, 0.0, b"", bytes([9]))], "residuals"),
    ([(2, 0, 40, bytes([5]), 0.0, 0.0, bytes([1, 2, 3, 4, 0]), b"")], "bitstream"),
])
def test_malformed_segments_are_errors(hip, rows, message):
    # The reference panics on these (models/mod.rs:170,237 etc.); the C ABI returns an error.
    batch = mdb.SegmentBatch.from_rows(rows)
    with pytest.raises(mdb.HipError, match=message):
        hip.grid_batch(batch, cap=64)
    with pytest.raises(ora.OracleError):
        ora.grid_batch(batch)


def test_output_capacity_is_checked(hip):
    batch = mdb.SegmentBatch.from_rows([(0, 0, 90, bytes([10]), 2.5, 2.5, b"", b"")])
    with pytest.raises(mdb.HipError, match="too small"):
        hip.grid_batch(batch, cap=4)


def test_grid_full_size_properties(hip):
    """Size-independent properties on a large batch: counts add up, timestamps strictly increase
    inside every series, every value within epsilon of the raw data."""
    eb = cases.error_bounds()["rel1"]
    n_series, n_points = 16, 200_000
    timestamps = np.arange(n_points, dtype=np.int64) * 1000
    all_values, parts = [], []
    import datagen
    for s in range(n_series):
        _, values = datagen.sine_series(s, n_points)
        all_values.append(values)
        parts.append(ora.try_compress_univariate_time_series(timestamps, values, eb))
    batch = mdb.SegmentBatch.concat(parts)
    ts, values, rows, metrics = hip.grid_batch(batch)
    assert len(ts) == n_series * n_points == int(rows.sum()) == metrics["rows_created"]
    ts = ts.reshape(n_series, n_points)
    assert (ts == timestamps[None, :]).all()
    raw = np.stack(all_values)
    approx = values.reshape(n_series, n_points)
    relative = np.abs((raw - approx) / raw) * np.float32(100.0)
    assert (relative <= np.float32(1.0)).all()


def test_config1_one_series_one_million_points_lossless(hip, monkeypatch):
    # BASELINE configs[0]: 1 univariate series, 1M regular-timestamp f32 points, lossless,
    # compress + grid. Chunked like the reference server (65 536 point buffers).
    import datagen
    n = 1_000_000
    timestamps, values = datagen.sine_series(0, n)
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
    offsets[-1] = n
    expected_segments = ora.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    segments = hip.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    assert segments.rows() == expected_segments.rows()
    assert len(segments) == 16 and set(segments.model_type_id.tolist()) == {2}
    ts, reconstructed, rows, metrics = hip.grid_batch(segments)
    assert np.array_equal(ts, timestamps)
    assert np.array_equal(reconstructed.view(np.uint32), values.view(np.uint32))  # lossless
    assert metrics["rows_created_by_macaque_v"] == n
    # 16 streams of 65 536 values from the host: the call's host threads walk them and leave cursors, every piece
    # of 64 values is then decoded by a lane of its own (k_grid_mv_pieces); without that index
    # (MDB_GRID_MV_INDEX=0) the parallel MacaqueV decoder takes them unless it is switched off (k_mv_decode
    # ran and decoded them: the one-lane decoder then has nothing left and the values came from it).
    for index in ("on", "off"):
        if index == "off":
            monkeypatch.setenv("MDB_GRID_MV_INDEX", "0")
        hip.profile_enable(True)
        hip.profile_reset()
        again = hip.grid_batch(segments)
        kernels = hip.profile()
        hip.profile_enable(False)
        assert np.array_equal(again[1].view(np.uint32), values.view(np.uint32))
        assert ("k_grid_mv_pieces" in kernels) == (index == "on")
        assert "k_mv_decode" in kernels and "k_mv_walk" in kernels
    monkeypatch.delenv("MDB_GRID_MV_INDEX")
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    state, reference = hip.agg_batch(segments, mask), ora.agg_batch(segments, mask)
    assert (state.count, state.min, state.max) == (n, values.min(), values.max())
    assert abs(state.sum - reference.sum) <= 1e-5 * abs(reference.sum)


@pytest.mark.parametrize("streams", [40, 130])
def test_many_long_lossless_streams(hip, streams, macaque_decoder, monkeypatch):
    # The parallel MacaqueV decoder stages half (above 12 288 pieces of 4 096 bits) or a quarter (above
    # 49 152) of each piece in LDS instead of all of it; 40 and 130 streams of 65 536 values are 15 600
    # and 50 700 pieces. Still bit for bit what one lane per stream produces. (Without the cursors of the call's
    # host threads, which would take such streams away from it.)
    import datagen
    monkeypatch.setenv("MDB_GRID_MV_INDEX", "0")
    n = 65536
    distinct = [datagen.sine_series(100 + k, n)[1] for k in range(5)]
    values = np.concatenate([distinct[k % 5] for k in range(streams)])
    timestamps = np.tile(np.arange(n, dtype=np.int64) * 1000, streams)
    offsets = (np.arange(streams + 1, dtype=np.uint64) * n)
    segments = hip.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    assert len(segments) == streams and set(segments.model_type_id.tolist()) == {2}
    hip.profile_enable(True)
    hip.profile_reset()
    ts, reconstructed, rows, metrics = hip.grid_batch(segments)
    kernels = hip.profile()
    hip.profile_enable(False)
    assert "k_mv_decode" in kernels  # every mode of this file's fixture leaves the decoder on for such streams
    assert np.array_equal(ts, timestamps)
    assert np.array_equal(reconstructed.view(np.uint32), values.view(np.uint32))
    assert metrics["rows_created_by_macaque_v"] == streams * n
    t_lo, t_hi = 20_000_000, 40_000_000
    ts, reconstructed, rows, _ = hip.grid_batch_range(segments, t_lo, t_hi)
    keep = (timestamps >= t_lo) & (timestamps <= t_hi)
    assert np.array_equal(ts, timestamps[keep])
    assert np.array_equal(reconstructed.view(np.uint32), values[keep].view(np.uint32))


def test_random_bit_patterns_survive_lossless_round_trip(hip):
    # Any f32 bit pattern (NaN payloads, subnormals, infinities) must come back bit for bit
    # (the reference's proptests use ProptestValue::ANY: macaque_v.rs:436-475, pmc_mean.rs:141-152).
    rng = np.random.default_rng(101)
    for n in (1, 2, 3, 9, 64, 300, 5000):
        values = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        timestamps = np.arange(n, dtype=np.int64) * 100
        segments = hip.try_compress_univariate_time_series(timestamps, values, cases.LOSSLESS)
        assert segments.rows()[0][:3] == ora.try_compress_univariate_time_series(
            timestamps, values, cases.LOSSLESS).rows()[0][:3]
        ts, reconstructed, _, _ = hip.grid_batch(segments)
        assert np.array_equal(ts, timestamps)
        got, want = reconstructed.view(np.uint32), values.view(np.uint32)
        nan = np.isnan(values)
        assert np.array_equal(got[~nan], want[~nan])
        assert np.isnan(reconstructed[nan]).all()   # NaNs stay NaNs (PMC/Swing may canonicalise payloads)


def test_values_only_grid_for_joined_field_columns(hip):
    # N3: two field columns of the same series share their timestamps; the second field's grid
    # skips the timestamp stores (out_ts = NULL).
    import datagen
    eb = cases.error_bounds()["rel1"]
    n = 50_000
    timestamps = np.arange(n, dtype=np.int64) * 1000
    fields = [ora.try_compress_univariate_time_series(timestamps, datagen.sine_series(s, n)[1], eb)
              for s in (3, 4)]
    devices = [hip.upload_segments(f) for f in fields]
    out_ts = hip.dev_alloc(8 * n)
    out_vals = [hip.dev_alloc(4 * n) for _ in fields]
    assert hip.grid_batch_dev(devices[0], out_ts, out_vals[0], n)[0] == n
    assert hip.grid_batch_dev(devices[1], None, out_vals[1], n)[0] == n
    assert np.array_equal(hip.download_array(out_ts, n, np.int64), timestamps)
    for field, pointer in zip(fields, out_vals):
        expected = ora.grid_batch(field)[1]
        assert np.array_equal(hip.download_array(pointer, n, np.float32).view(np.uint32),
                              expected.view(np.uint32))
    for pointer in [out_ts] + out_vals:
        hip.dev_free(pointer)


def _expected_range(batch, t_lo, t_hi):
    ts, values, rows, _ = ora.grid_batch(batch)
    keep = (ts >= t_lo) & (ts <= t_hi)
    starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    per_segment = np.array([int(keep[int(starts[k]):int(starts[k + 1])].sum()) for k in range(len(rows))],
                           dtype=np.uint32)
    return ts[keep], values[keep], per_segment


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("eb_name", ["lossless", "rel5", "abs5"])
def test_grid_with_pushed_down_time_range(hip, eb_name, irregular):
    # GridStream reconstructs everything and prunes afterwards (grid_exec.rs:366-387); the pushed
    # down range must give the same rows in the same order.
    eb = cases.error_bounds()[eb_name]
    timestamps, _, batch = cases.mixed_batch(eb, irregular, seed=121)
    n = len(timestamps)
    windows = [
        (int(timestamps[n // 4]), int(timestamps[3 * n // 4])),
        (int(timestamps[0]), int(timestamps[-1])),
        (int(timestamps[100]), int(timestamps[100])),
        (int(timestamps[-1]) + 1, int(timestamps[-1]) + 1000),        # nothing
        (-(1 << 62), 1 << 62),                                        # everything
        (int(timestamps[n // 2]) - 37, int(timestamps[n // 2]) + 4242),
        (int(timestamps[17]) + 1, int(timestamps[5000]) - 1),
    ]
    for t_lo, t_hi in windows:
        exp_ts, exp_values, exp_rows = _expected_range(batch, t_lo, t_hi)
        ts, values, rows, metrics = hip.grid_batch_range(batch, t_lo, t_hi)
        assert np.array_equal(ts, exp_ts), (t_lo, t_hi)
        assert np.array_equal(values.view(np.uint32), exp_values.view(np.uint32)), (t_lo, t_hi)
        assert np.array_equal(rows, exp_rows)
        assert metrics["rows_created"] == len(exp_ts)


@pytest.mark.parametrize("points_per_series, n_series", [(9_000, 12), (250_000, 4)])
def test_time_range_with_runs_of_segments_without_a_visible_point(hip, points_per_series, n_series):
    """A query over a thin slice of the time axis of several series (BASELINE config 5's shape): between the visible
    parts of two series lie the segments behind the range of the one and in front of the range of the other - about a
    thousand, and about thirty thousand (more than the tile kernel's table of offsets holds) - inside ONE output tile,
    which keeps the handful of segments that have points in it."""
    eb = mdb.error_bound("absolute", 0.25)
    timestamps = np.arange(points_per_series, dtype=np.int64) * 1000
    parts = []
    for series in range(n_series):
        # a level every 8 points: segments of 8 points (PMC-Mean), with a linear stretch (Swing) now and then
        levels = np.repeat(np.arange(points_per_series // 8 + 1) % 7 * 3.0 + series, 8)[:points_per_series]
        values = levels.astype(np.float32)
        values[1000:1400] = np.linspace(0.0, 50.0, 400, dtype=np.float32)
        parts.append(ora.try_compress_univariate_time_series(timestamps, values, eb))
    batch = mdb.SegmentBatch.concat(parts)
    assert len(batch) > n_series * points_per_series // 10
    n = points_per_series
    for first, last in ((n // 2, n // 2 + n // 20), (1100, 1300), (n - 300, n - 1), (3, 5)):
        t_lo, t_hi = int(timestamps[first]) - 1, int(timestamps[last]) + 1
        exp_ts, exp_values, exp_rows = _expected_range(batch, t_lo, t_hi)
        ts, values, rows, metrics = hip.grid_batch_range(batch, t_lo, t_hi)
        assert len(exp_ts) == n_series * (last - first + 1)
        assert np.array_equal(ts, exp_ts), (first, last)
        assert np.array_equal(values.view(np.uint32), exp_values.view(np.uint32)), (first, last)
        assert np.array_equal(rows, exp_rows)


@pytest.mark.parametrize("gap_probability", [0.0005, 0.005, 0.01, 0.02, 0.5])
def test_mostly_regular_timestamps_with_gaps(hip, gap_probability):
    # What irregular timestamps usually are: a fixed sampling interval with a sample missing now and
    # then. One gap makes a whole segment irregular (timestamps.rs:77-96) and its stream then consists
    # of long runs of `0` codes, which the decoder takes off the stream up to 32 at a time.
    rng = np.random.default_rng(171)
    n = 60_000
    timestamps = 1_700_000_000_000_000 + np.cumsum(np.where(rng.random(n) < gap_probability, 3000, 1000).astype(np.int64))
    values = (50 + 5 * np.sin(np.arange(n) / 700.0) + rng.uniform(-0.02, 0.02, n)).astype(np.float32)
    offsets = np.array([0, 7, 7 + 40_000, n], dtype=np.uint64)
    for eb_name in ("rel1", "lossless"):
        eb = cases.error_bounds()[eb_name]
        expected_segments = ora.compress_chunks(timestamps, values, offsets, eb)
        segments = hip.compress_chunks(timestamps, values, offsets, eb)
        assert segments.rows() == expected_segments.rows()
        cases.assert_grid_equal(hip.grid_batch(segments), ora.grid_batch(segments))
        assert np.array_equal(hip.grid_batch(segments)[0], timestamps)
        for t_lo, t_hi in ((int(timestamps[1000]), int(timestamps[1031])), (int(timestamps[33]) + 1, int(timestamps[50_000]) - 1),
                           (int(timestamps[-1]), int(timestamps[-1]) + 5)):
            exp_ts, exp_values, exp_rows = _expected_range(segments, t_lo, t_hi)
            ts, reconstructed, rows, _ = hip.grid_batch_range(segments, t_lo, t_hi)
            assert np.array_equal(ts, exp_ts), (t_lo, t_hi)
            assert np.array_equal(reconstructed.view(np.uint32), exp_values.view(np.uint32)), (t_lo, t_hi)
            assert np.array_equal(rows, exp_rows)
        mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
        got, expected = hip.agg_batch(segments, mask), ora.agg_batch(segments, mask)
        assert (got.count, got.min, got.max) == (expected.count, expected.min, expected.max)
        assert abs(got.sum - expected.sum) <= 1e-5 * abs(expected.sum)


def test_fixed_rate_series_with_a_few_very_long_gaps(hip):
    # Long runs of `0` codes are written as arithmetic runs by the whole wave (k_grid_timestamps), staged as
    # 32-bit distances and deltas: a gap of hours (a delta beyond 2^32 microseconds) or a piece that starts
    # more than 2^32 microseconds behind its stream's checkpoint must fall back to stores of 64-bit values.
    rng = np.random.default_rng(172)
    n = 200_000
    deltas = np.full(n, 1000, dtype=np.int64)
    deltas[rng.integers(1000, n - 1000, 6)] = 7_200_000_000        # two hours
    deltas[rng.integers(1000, n - 1000, 40)] = 4_294_967_296 + 7   # just beyond 32 bits
    deltas[rng.integers(1000, n - 1000, 200)] = 5000
    timestamps = 1_600_000_000_000_000 + np.cumsum(deltas)
    values = (20 + 3 * np.sin(np.arange(n) / 900.0) + rng.uniform(-0.01, 0.01, n)).astype(np.float32)
    offsets = np.arange(0, n + 1, 50_000, dtype=np.uint64)
    for eb_name in ("rel1", "lossless"):
        segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
        got = hip.grid_batch(segments)
        cases.assert_grid_equal(got, ora.grid_batch(segments))
        assert np.array_equal(got[0], timestamps)
        t_lo, t_hi = int(timestamps[70_001]), int(timestamps[160_000])
        exp_ts, exp_values, exp_rows = _expected_range(segments, t_lo, t_hi)
        ts, reconstructed, rows, _ = hip.grid_batch_range(segments, t_lo, t_hi)
        assert np.array_equal(ts, exp_ts) and np.array_equal(rows, exp_rows)
        assert np.array_equal(reconstructed.view(np.uint32), exp_values.view(np.uint32))


def test_jump_lists_at_their_edges(hip):
    # A fixed rate with the odd gap is written by k_grid_tiles from the segment's jump list (TsJump): the first
    # delta is what every other one is measured against; a list holds one jump per 64 bits of the stream and
    # at most one per sixteen points; the last deltas of a stream are found by the careful decoder and the
    # very last point is end_time, not a code. Every shape here sits on one of those edges, and segments of
    # both kinds (with a list, decoded piece by piece) lie next to each other in one batch.
    rng = np.random.default_rng(173)

    def series(n, gaps_at, gap=3000, interval=1000):
        deltas = np.full(n, interval, dtype=np.int64)
        deltas[np.asarray(gaps_at, dtype=np.int64)] = gap
        return 1_650_000_000_000_000 + np.cumsum(deltas)

    n = 4000
    shapes = {
        "no-gap-but-one": series(n, [n // 2]),
        "first-delta-is-the-gap": series(n, [1]),
        "second-delta-is-the-gap": series(n, [2]),
        "gap-before-the-last-point": series(n, [n - 1]),
        "gaps-in-the-last-points": series(n, [n - 6, n - 4, n - 3, n - 2, n - 1]),
        "a-sample-early-and-one-late": series(n, [100, 101, 2000], gap=1001) - np.where(np.arange(n) >= 3000, 1, 0),
        "sixteen-gaps-in-a-row": series(n, list(range(500, 516))),
        "a-gap-every-15": series(n, list(range(20, n, 15))),
        "a-gap-every-17": series(n, list(range(20, n, 17))),
        "a-gap-every-40": series(n, list(range(20, n, 40))),
        "hours": series(n, [700, 701, 3000], gap=7_200_000_000),
        "beyond-32-bits": series(n, [9, 1999, n - 2], gap=4_294_967_296 + 7),
        "one-in-a-hundred": 1_650_000_000_000_000 + np.cumsum(np.where(rng.random(n) < 0.01, 2000, 1000).astype(np.int64)),
    }
    for name, timestamps in shapes.items():
        assert np.all(np.diff(timestamps) > 0), name
    timestamps = np.concatenate(list(shapes.values()))
    offsets = np.arange(0, len(timestamps) + 1, n, dtype=np.uint64)
    smooth = (30 + 4 * np.sin(np.arange(len(timestamps)) / 300.0)).astype(np.float32)
    noisy = (smooth + rng.uniform(-2, 2, len(timestamps)).astype(np.float32)).astype(np.float32)
    for values, eb_name in ((smooth, "rel1"), (smooth, "lossless"), (noisy, "rel1"), (noisy, "abs5")):
        segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
        expected = ora.grid_batch(segments)
        got = hip.grid_batch(segments)
        cases.assert_grid_equal(got, expected)
        assert np.array_equal(got[0], timestamps)
        # a joined field column: values only, the Swing ones computed from timestamps that are not stored
        _, only_values, _, _ = hip.grid_batch_owned(segments, values_only=True)
        assert np.array_equal(only_values.view(np.uint32), expected[1].view(np.uint32))
        # segment by segment, and the segments in another order (a list belongs to its segment, not its place)
        order = rng.permutation(len(segments))
        shuffled = segments.take(order)
        cases.assert_grid_equal(hip.grid_batch(shuffled), ora.grid_batch(shuffled))
        t_lo, t_hi = int(timestamps[n + 1990]), int(timestamps[9 * n + 30])
        exp_ts, exp_values, exp_rows = _expected_range(segments, t_lo, t_hi)
        ts, reconstructed, rows, _ = hip.grid_batch_range(segments, t_lo, t_hi)
        assert np.array_equal(ts, exp_ts) and np.array_equal(rows, exp_rows)
        assert np.array_equal(reconstructed.view(np.uint32), exp_values.view(np.uint32))


def test_randomly_spaced_and_gapped_series_in_one_batch(hip):
    # The pieces that are left to k_grid_timestamps are listed only by waves of the counting walk that have met a
    # segment with a jump list; the list is used if it turns out complete. Here it does not (the walk takes the
    # streams longest first: its first waves see randomly spaced timestamps only), in a second batch it does, and
    # in a third nothing is left to list.
    rng = np.random.default_rng(174)
    n_random, n_gapped, chunk = 80_000, 40_000, 500
    random_ts = np.cumsum(rng.integers(900, 1100, n_random).astype(np.int64))
    gapped_ts = np.cumsum(np.where(rng.random(n_gapped) < 0.01, 2000, 1000).astype(np.int64))
    for n_a, n_b in ((n_random, n_gapped), (3 * chunk, n_gapped), (0, n_gapped)):
        timestamps = np.concatenate([1_640_000_000_000_000 + random_ts[:n_a],
                                     1_640_000_000_000_000 + (int(random_ts[n_a - 1]) if n_a else 0) + gapped_ts[:n_b]])
        values = (10 + np.sin(np.arange(len(timestamps)) / 200.0)).astype(np.float32)
        offsets = np.arange(0, len(timestamps) + 1, chunk, dtype=np.uint64)
        for eb_name in ("rel1", "lossless"):
            segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
            got = hip.grid_batch(segments)
            cases.assert_grid_equal(got, ora.grid_batch(segments))
            assert np.array_equal(got[0], timestamps)


def test_hundreds_of_listed_segments_in_one_tile(hip):
    # A tile of k_grid_tiles_jumps with more than 128 segments (their descriptors are not all in LDS) or more rows
    # than its table holds is done without the table: the waves read the lists of their points' segments
    # together. Tiny chunks whose one gap is hours long have streams long enough (a 69-bit code each way) to get
    # checkpoints and a list: four to sixteen points per segment, up to a thousand listed segments per tile.
    rng = np.random.default_rng(175)
    for chunk in (4, 5, 8, 16):
        n = chunk * 3000
        deltas = np.full(n, 1000, dtype=np.int64)
        deltas[np.arange(2, n, chunk)] = 8_000_000_000_000 + rng.integers(0, 1000, len(np.arange(2, n, chunk)))
        timestamps = 1_500_000_000_000_000 + np.cumsum(deltas)
        values = np.repeat(rng.normal(5, 1, n // chunk).astype(np.float32), chunk)   # PMC-Mean
        values[n // 2:] += (np.arange(n - n // 2) % chunk).astype(np.float32) * 0.5     # Swing
        offsets = np.arange(0, n + 1, chunk, dtype=np.uint64)
        for eb_name in ("rel1", "lossless"):
            segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
            got = hip.grid_batch(segments)
            cases.assert_grid_equal(got, ora.grid_batch(segments))
            assert np.array_equal(got[0], timestamps)


def test_jump_lists_longer_than_a_wave(hip):
    # k_grid_tiles reads a segment's jump list 64 entries at a time, the lanes of the wave that lie in the
    # segment together; a list of more than 64 entries is first narrowed down by probes of the whole wave (of
    # more than 4 096: twice). Constant values make segments as long as their chunks.
    for n, every in ((20_000, 40), (60_000, 100), (300_000, 50), (300_000, 17)):
        deltas = np.full(n, 1000, dtype=np.int64)
        deltas[np.arange(30, n, every)] = 2000
        deltas[n // 3] = 3_000_000_000_000
        timestamps = 1_650_000_000_000_000 + np.cumsum(deltas)
        values = np.full(n, 7.25, dtype=np.float32)
        values[n // 2:] = np.linspace(1.0, 2.0, n - n // 2, dtype=np.float32)   # (a line: Swing, if it holds)
        offsets = np.array([0, n], dtype=np.uint64)
        for eb_name in ("rel1", "lossless"):
            segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
            got = hip.grid_batch(segments)
            cases.assert_grid_equal(got, ora.grid_batch(segments))
            assert np.array_equal(got[0], timestamps)


def test_grid_time_range_edge_cases(hip):
    for eb_name in ("lossless", "rel5"):
        batch = cases.edge_case_batch(cases.error_bounds()[eb_name])
        all_ts = ora.grid_batch(batch)[0]
        for t_lo, t_hi in ((0, 0), (100, 450), (1000, 1700), (250, 27_000), (1658671178037, 1658671200000),
                           (int(all_ts.min()), int(all_ts.max())), (5, 6)):
            exp_ts, exp_values, exp_rows = _expected_range(batch, t_lo, t_hi)
            ts, values, rows, _ = hip.grid_batch_range(batch, t_lo, t_hi)
            assert np.array_equal(ts, exp_ts), (eb_name, t_lo, t_hi)
            assert np.array_equal(values.view(np.uint32), exp_values.view(np.uint32)), (eb_name, t_lo, t_hi)
            assert np.array_equal(rows, exp_rows)


def test_views_that_point_outside_their_buffers_are_errors(hip):
    """A BinaryView whose buffer index or offset leaves the column's data buffers must be an error
    from every host entry point (the kernels follow views without looking); arrow cannot build such
    a column, a foreign or corrupted batch can."""
    import ctypes as C
    timestamps, values = cases.synthetic_series(4000, True, None, seed=77)
    good = ora.try_compress_univariate_time_series(timestamps, values, cases.LOSSLESS)
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_SUM
    assert len(hip.grid_batch(good)[0]) == 4000

    def corrupted(column_name, field, value):
        batch = good.take(np.arange(len(good)))
        column = getattr(batch, column_name)
        lengths = column.views[:, 0:4].copy().view(np.int32).reshape(-1)
        row = int(np.nonzero(lengths > 12)[0][0])
        column.views[row, field:field + 4] = np.frombuffer(np.int32(value).tobytes(), dtype=np.uint8)
        return batch

    rng = np.random.default_rng(9)
    for column_name in ("timestamps", "values"):
        size = len(getattr(good, column_name).buffers[0])
        for field, value in ((8, 1), (8, 7), (8, -1), (12, -4), (12, size - 3), (12, size + 100),
                             (12, 2**31 - 1), (0, -5), (8, int(rng.integers(2, 2**31 - 1)))):
            bad = corrupted(column_name, field, value)
            for call in (lambda: hip.grid_batch(bad), lambda: hip.grid_count(bad),
                         lambda: hip.grid_batch_owned(bad), lambda: hip.agg_batch(bad, mask),
                         lambda: hip.agg_batch_range(bad, 0, 10**9, mask), lambda: hip.upload_segments(bad)):
                with pytest.raises(mdb.HipError, match="Malformed BinaryView|negative"):
                    call()
    # a device batch that did not come from this library: mdb_segments_validate_dev finds the same
    dev = hip.upload_segments(good)
    hip.validate_segments_dev(dev)
    views = good.values.views.copy()
    lengths = views[:, 0:4].copy().view(np.int32).reshape(-1)
    row = int(np.nonzero(lengths > 12)[0][0])
    views[row, 12:16] = np.frombuffer(np.int32(len(good.values.buffers[0]) - 1).tobytes(), dtype=np.uint8)
    address = C.cast(dev.seg.values.views, C.c_void_p).value
    hip._check(hip.lib.mdb_dev_upload(hip.handle, C.c_void_p(address), views.ctypes.data_as(C.c_void_p), views.nbytes))
    with pytest.raises(mdb.HipError, match="Malformed BinaryView"):
        hip.validate_segments_dev(dev)
    dev.free()
    assert len(hip.grid_batch(good)[0]) == 4000  # the context is fine afterwards


def test_fuzzed_segments_never_crash_and_agree_with_the_oracle(hip):
    """Valid segments with random corruptions (truncated or random payloads, swapped times, wrong
    model type): the reference would panic on many of them; the C ABI must return an error or a
    result, never crash or hang, and whenever the oracle accepts a batch the GPU must accept it too
    and produce the same points."""
    rng = np.random.default_rng(131)
    pool = []
    for eb_name in ("lossless", "rel5"):
        pool += cases.edge_case_batch(cases.error_bounds()[eb_name]).rows()
        for irregular in (False, True):
            pool += cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=132, length=3000)[2].rows()
    agree = errors = 0
    for trial in range(400):
        rows = []
        for _ in range(int(rng.integers(1, 6))):
            row = list(pool[int(rng.integers(0, len(pool)))])
            if rng.random() < 0.35:
                field = int(rng.choice([0, 1, 2, 3, 6, 7]))
                if field == 0:
                    row[0] = int(rng.integers(0, 4))
                elif field in (1, 2):
                    row[field] = int(row[field] + rng.integers(-500, 500))
                else:
                    payload = bytearray(row[field])
                    action = rng.integers(0, 3)
                    if action == 0 and payload:
                        payload = payload[: int(rng.integers(0, len(payload)))]
                    elif action == 1 and payload:
                        payload[int(rng.integers(0, len(payload)))] ^= 1 << int(rng.integers(0, 8))
                    else:
                        payload = bytearray(rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8).tobytes())
                    row[field] = bytes(payload)
            rows.append(tuple(row))
        batch = mdb.SegmentBatch.from_rows(rows)
        try:
            expected = ora.grid_batch(batch)
        except ora.OracleError:
            expected = None
        if expected is not None and len(expected[0]) > 200_000:
            continue  # a corrupted length can ask for an absurd output; not the point of this test
        try:
            got = hip.grid_batch(batch, cap=200_000)
        except mdb.HipError:
            got = None
        if expected is not None:
            assert got is not None, rows
            cases.assert_grid_equal(got, expected)
            agree += 1
        else:
            errors += got is None
    assert agree > 50 and errors > 20, (agree, errors)


def test_owned_grid_results_in_page_locked_memory(hip):
    eb = cases.error_bounds()["rel5"]
    timestamps, _, batch = cases.mixed_batch(eb, True, seed=141)
    expected = ora.grid_batch(batch)
    for _ in range(3):   # the second and third call reuse the pooled page-locked block
        got = hip.grid_batch_owned(batch)
        cases.assert_grid_equal(got, expected)
        assert np.array_equal(got[2], expected[2]) and got[3] == expected[3]
    lo, hi = int(timestamps[500]), int(timestamps[9000])
    ranged = hip.grid_batch_owned(batch, time_range=(lo, hi))
    cases.assert_grid_equal(ranged, hip.grid_batch_range(batch, lo, hi))
    ts, values, rows, metrics, release = hip.grid_batch_owned(batch, copy=False)
    assert np.array_equal(ts, expected[0])
    release()
    empty = hip.grid_batch_owned(mdb.SegmentBatch.from_rows([]))
    assert len(empty[0]) == 0 and len(empty[2]) == 0


def test_trim_gives_memory_back_and_the_context_keeps_working(hip):
    eb = cases.error_bounds()["rel5"]
    _, _, batch = cases.mixed_batch(eb, False, seed=31)
    before = hip.grid_batch(batch)
    owned = hip.grid_batch_owned(batch, copy=False)  # a result handed out before the trim stays valid
    state = hip.agg_batch(batch, mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT)
    assert hip.trim() > 0
    assert hip.trim() == 0
    assert np.array_equal(owned[1].view(np.uint32), before[1].view(np.uint32))
    owned[4]()
    after = hip.grid_batch(batch)
    assert np.array_equal(after[0], before[0])
    assert np.array_equal(after[1].view(np.uint32), before[1].view(np.uint32))
    again = hip.agg_batch(batch, mdb.MDB_AGG_SUM | mdb.MDB_AGG_COUNT)
    assert (again.sum, again.count) == (state.sum, state.count)


def test_concurrent_calls_from_several_threads(hip):
    # DataFusion polls GridStreams from tokio worker threads (SURVEY 8(b) "Threading"): calls on ONE
    # context are serialised inside the library, separate contexts run side by side. ctypes releases
    # the GIL during the calls, so these threads really overlap.
    import threading
    eb = cases.error_bounds()["rel5"]
    batches = [cases.mixed_batch(eb, irregular, seed=140 + k, length=8000)[2] for k, irregular in
               enumerate((False, True, False, True))]
    expected = [ora.grid_batch(b) for b in batches]
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    own_contexts = [mdb.Context(0) for _ in batches]
    failures = []

    def worker(k, context):
        try:
            for _ in range(20):
                got = context.grid_batch(batches[k])
                cases.assert_grid_equal(got, expected[k])
                assert context.agg_batch(batches[k], mask).count == len(expected[k][0])
        except Exception as error:  # noqa: BLE001 - reported below
            failures.append((k, repr(error)))

    for contexts in ([hip] * len(batches), own_contexts):
        threads = [threading.Thread(target=worker, args=(k, contexts[k])) for k in range(len(batches))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "a call did not return"
        assert not failures, failures
    for context in own_contexts:
        context.close()


def test_scratch_limit_gives_memory_back_after_every_call():
    # mdb_set_scratch_limit: a context keeps at most that much device scratch between calls (an owner of
    # many contexts - one GridStream per field column - would otherwise keep what each one's largest
    # batch needed until it calls mdb_trim). Results are what they are without the limit.
    from modelardb_rs_amd import api
    timestamps, values = cases.synthetic_series(400_000, False, (1.0, 1.05), seed=5)
    batch = ora.try_compress_univariate_time_series(timestamps, values, cases.error_bounds()["rel1"])
    expected = ora.grid_batch(batch)
    unlimited, limited = api.Context(0), api.Context(0)
    limited.set_scratch_limit(1 << 20)
    for context in (unlimited, limited):
        for _ in range(2):
            cases.assert_grid_equal(context.grid_batch(batch), expected)
    kept_unlimited, kept_limited = unlimited.trim(), limited.trim()
    assert kept_unlimited > 4 * len(timestamps)          # the output staging alone is 12 bytes per point
    assert kept_limited <= 1 << 20
    limited.set_scratch_limit(0)
    cases.assert_grid_equal(limited.grid_batch(batch), expected)
    assert limited.trim() > 4 * len(timestamps)
    unlimited.close()
    limited.close()


def test_clones_are_kept_for_the_next_clone_and_survive_their_origin():
    # mdb_clone / mdb_close: a closed clone (stream, scratch) waits for the next mdb_clone of the same context;
    # the origin's close destroys the waiting ones; a clone still in use then is simply closed when its time comes.
    from modelardb_rs_amd import api
    timestamps, values = cases.synthetic_series(50_000, True, (1.0, 1.05), seed=8)
    batch = ora.try_compress_univariate_time_series(timestamps, values, cases.error_bounds()["rel1"])
    expected = ora.grid_batch(batch)
    origin = api.Context(0)
    first = origin.clone()
    cases.assert_grid_equal(first.grid_batch(batch), expected)
    kept_handle = first.handle.value
    first.close()
    second, third = origin.clone(), origin.clone()
    assert second.handle.value == kept_handle and third.handle.value != kept_handle   # recycled, then a new one
    for context in (second, third, origin):
        cases.assert_grid_equal(context.grid_batch(batch), expected)
    second.close()
    origin.close()                                   # destroys the idle clone; `third` is still in use
    cases.assert_grid_equal(third.grid_batch(batch), expected)
    third.close()


def _tag_views(rng, n, long_share=0.3, buffers=2):
    """n random Utf8View views the way arrow lays them out (length, then the bytes or prefix / buffer / offset)."""
    views = np.zeros((n, 16), dtype=np.uint8)
    lengths = np.where(rng.random(n) < long_share, rng.integers(13, 40, n), rng.integers(0, 13, n)).astype("<i4")
    views[:, 0:4] = lengths.view(np.uint8).reshape(-1, 4)
    views[:, 4:16] = rng.integers(0, 256, (n, 12), dtype=np.uint8)
    long_rows = lengths > 12
    views[long_rows, 8:12] = rng.integers(0, buffers, int(long_rows.sum())).astype("<i4").view(np.uint8).reshape(-1, 4)
    short = ~long_rows
    for k in range(12):  # inline bytes beyond the length are zero
        views[short & (lengths <= k), 4 + k] = 0
    return views, long_rows


def _expected_tags(views, long_rows, rows, shift):
    expected = np.repeat(views, rows, axis=0)
    index = expected[:, 8:12].copy().view("<i4").reshape(-1)
    index[np.repeat(long_rows, rows)] += shift
    expected[:, 8:12] = index.view(np.uint8).reshape(-1, 4)
    return expected


def test_submitted_batches_come_back_as_one_with_their_tags_repeated(hip):
    # mdb_grid_submit / mdb_grid_wait (what the patched GridStream::poll_next calls, grid_exec.rs:261-429): several
    # input batches through one launch, rows in the order of the list; tag views repeated per created row
    # (grid_exec.rs:339-346), long ones moved onto the output column's buffer list.
    rng = np.random.default_rng(17)
    eb = cases.error_bounds()["rel1"]
    batches = [cases.mixed_batch(eb, irregular, seed=300 + k, length=length)[2]
               for k, (irregular, length) in enumerate(((False, 9000), (True, 7000), (False, 0), (False, 12_000)))]
    batches[2] = mdb.SegmentBatch.from_rows([])                       # an empty input in the middle
    batches.append(cases.edge_case_batch())                           # MacaqueV streams, residual tails, data buffers
    joined = mdb.SegmentBatch.concat(batches)
    expected = ora.grid_batch(joined)
    tags = [[_tag_views(rng, len(batch)) for _ in range(2)] for batch in batches]
    shifts = [[1 + 2 * b, 5 + 3 * b] for b in range(len(batches))]
    for reserve_front in (0, 37, 8192):
        ticket = hip.grid_submit(batches, [[views for views, _ in columns] for columns in tags], shifts,
                                 reserve_front=reserve_front)
        ts, values, rows, metrics, tag_columns = ticket.wait()
        cases.assert_grid_equal((ts, values), expected)
        assert np.array_equal(rows, expected[2]) and metrics == expected[3]
        at_segment, at_row = 0, 0
        for b, batch in enumerate(batches):
            batch_rows = rows[at_segment:at_segment + len(batch)]
            created = int(batch_rows.sum())
            for column in range(2):
                views, long_rows = tags[b][column]
                assert np.array_equal(tag_columns[column][at_row:at_row + created],
                                      _expected_tags(views, long_rows, batch_rows, shifts[b][column]))
            at_segment, at_row = at_segment + len(batch), at_row + created
        assert at_row == len(ts)
    # one input: the same as mdb_grid_batch_owned, with and without a time range, values only
    lo, hi = int(expected[0][len(expected[0]) // 5]), int(expected[0][len(expected[0]) // 2])
    for batch in batches:
        plain = hip.grid_submit([batch]).wait()
        cases.assert_grid_equal(plain, ora.grid_batch(batch))
        ranged = hip.grid_submit([batch], time_range=(lo, hi)).wait()
        cases.assert_grid_equal(ranged, hip.grid_batch_range(batch, lo, hi))
        only_values = hip.grid_submit([batch], values_only=True).wait()
        assert only_values[0] is None
        assert np.array_equal(only_values[1].view(np.uint32), plain[1].view(np.uint32))
    ranged = hip.grid_submit(batches, time_range=(lo, hi)).wait()
    cases.assert_grid_equal(ranged, hip.grid_batch_range(joined, lo, hi))


def test_many_tickets_in_flight_waited_in_any_order_and_cancelled(hip):
    # Two submits run at once (the context and the clone the library keeps), further ones queue; results may be
    # waited for in any order; a ticket nobody waits for is cancelled (a GridStream dropped mid-query); a failing
    # batch fails its own wait only, with its own message on the waiting thread.
    eb = cases.error_bounds()["rel5"]
    batches = [cases.mixed_batch(eb, k % 2 == 1, seed=400 + k, length=3000 + 500 * k)[2] for k in range(7)]
    expected = [ora.grid_batch(batch) for batch in batches]
    bad = mdb.SegmentBatch.from_rows([(7, 100, 500, bytes([5]), 1.0, 1.0, b"", b"")])  # unknown model type
    from modelardb_rs_amd import api
    for context in (hip, api.Context(0)):
        for round_ in range(3):
            tickets = [context.grid_submit([batch]) for batch in batches]
            failing = context.grid_submit([bad])
            order = np.random.default_rng(round_).permutation(len(tickets))
            for k in order[:5]:
                cases.assert_grid_equal(tickets[k].wait(), expected[k])
            with pytest.raises(mdb.HipError, match="model type"):
                failing.wait()
            for k in order[5:]:
                tickets[k].cancel()
            cases.assert_grid_equal(context.grid_batch(batches[0]), expected[0])   # the context still works
    context.grid_submit([batches[1]])   # (an outstanding ticket when the context is closed: finished, then freed)
    context.close()


def test_a_reused_clone_is_a_fresh_one(hip):
    # mdb_close of a clone keeps it for the next mdb_clone - but not what its owner did to it: a stream set with
    # mdb_set_stream may be destroyed behind it (such a clone is closed for real), a communicator is closed,
    # timings are forgotten.
    from modelardb_rs_amd import api
    import torch
    eb = cases.error_bounds()["rel5"]
    _, _, batch = cases.mixed_batch(eb, False, seed=77, length=5000)
    expected = ora.grid_batch(batch)
    origin = api.Context(0)
    first = origin.clone()
    first.profile_enable(True)
    first.comm_init(0, 1, mdb.comm_unique_id())
    cases.assert_grid_equal(first.grid_batch(batch), expected)
    kept = first.handle.value
    first.close()
    second = origin.clone()
    assert second.handle.value == kept
    assert second.profile() == {}                      # nothing of the previous owner's launches
    second.comm_init(0, 1, mdb.comm_unique_id())       # ("already has a communicator" before)
    cases.assert_grid_equal(second.grid_batch(batch), expected)
    stream = torch.cuda.Stream()
    second.set_stream(stream.cuda_stream)
    cases.assert_grid_equal(second.grid_batch(batch), expected)
    second.close()                                     # not kept: its stream is the caller's
    del stream
    third = origin.clone()
    cases.assert_grid_equal(third.grid_batch(batch), expected)
    third.close()
    origin.close()


@pytest.mark.parametrize("index", ["cursors", "off"])
def test_resident_batches_are_decoded_piece_by_piece(hip, index, monkeypatch):
    # A batch that stays on the device gets, with its first grid call, a cursor in front of every 64th value of
    # every MacaqueV stream (k_mv_index_walk); from then on one lane decodes one piece (k_grid_mv_pieces) instead
    # of one lane one stream (macaque_v.rs:272-323 is sequential). Long lossless streams, residual tails of every
    # length up to 255 and beyond (a separate MacaqueV segment), streams inside their views, irregular timestamps,
    # specials; whole and under time ranges that begin and end inside pieces. MDB_GRID_MV_INDEX=0: the same
    # through the serial kernel.
    if index == "off":
        monkeypatch.setenv("MDB_GRID_MV_INDEX", "0")
    rng = np.random.default_rng(29)
    batches = [cases.edge_case_batch(), cases.edge_case_batch(cases.error_bounds()["rel5"])]
    for eb_name in ("lossless", "abs0.01", "rel1"):
        for irregular in (False, True):
            batches.append(cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=61, length=30_000)[2])
    noise = rng.uniform(-1e3, 1e3, 200_000).astype(np.float32)         # long streams: every value a new window
    offsets = np.arange(0, len(noise) + 1, 50_000, dtype=np.uint64)
    batches.append(ora.compress_chunks(np.arange(len(noise), dtype=np.int64) * 10, noise, offsets, cases.LOSSLESS))
    tails = []                                                          # residual tails of 1..300 values
    for n_res in (1, 2, 63, 64, 65, 127, 128, 129, 200, 255, 256, 300):
        values = np.concatenate([np.full(20, 5.0, dtype=np.float32), rng.uniform(-1e30, 1e30, n_res).astype(np.float32),
                                 np.arange(20, dtype=np.float32) * 3 + 1])
        tails.append(ora.try_compress_univariate_time_series(np.arange(len(values), dtype=np.int64) * 100, values, cases.LOSSLESS))
    batches.append(mdb.SegmentBatch.concat(tails))
    for batch in batches:
        expected = ora.grid_batch(batch)
        resident = hip.upload_segments(batch)
        for _ in range(2):   # (the first call builds the index)
            cases.assert_grid_equal(hip.grid_resident(resident), expected)
        all_ts = expected[0]
        for a, b in ((len(all_ts) // 7, len(all_ts) // 3), (0, 70), (len(all_ts) - 100, len(all_ts) - 1), (5, 5)):
            lo, hi = int(np.sort(all_ts)[a]), int(np.sort(all_ts)[min(b, len(all_ts) - 1)])
            keep = (all_ts >= lo) & (all_ts <= hi)
            got_ts, got_values = hip.grid_resident(resident, (lo, hi))
            assert np.array_equal(got_ts, all_ts[keep])
            assert np.array_equal(got_values.view(np.uint32), expected[1][keep].view(np.uint32))
        resident.free()
    # a batch the fitter leaves on the device is resident too
    values_dev = hip.upload_array(noise)
    offsets_dev = hip.upload_array(offsets)
    fitted = hip.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, cases.LOSSLESS, 0, 10,
                                     hip.upload_array(offsets[:-1] * 0))
    for _ in range(2):
        ts, values = hip.grid_resident(fitted)
        assert np.array_equal(values.view(np.uint32), noise.view(np.uint32))
    fitted.free()


def test_piece_decoder_beyond_two_to_the_31_values(hip):
    # Output positions are 64-bit everywhere: 45 000 lossless streams of 50 000 values are 2.25 x 10^9 values, past
    # 2^31 (the wave hands a row's output position from lane to lane in two 32-bit halves; a sign-extended low half
    # once wrote below the column). Sampled streams against the generator they were fitted from.
    import datagen
    streams, per = 45_000, 50_000
    n = streams * per
    values = hip.dev_alloc(4 * n)
    hip.synth_values_dev(values, 0, streams, per)
    offsets_dev = hip.upload_array(np.arange(0, n + per, per, dtype=np.uint64))
    fitted = hip.compress_chunks_dev(0, values, offsets_dev, streams, cases.LOSSLESS, 0, 1000, 0)
    hip.dev_free(values)
    assert hip.grid_count_dev(fitted) == n
    out_ts, out_val = hip.dev_alloc(8 * n), hip.dev_alloc(4 * n)
    for _ in range(2):
        produced, metrics = hip.grid_batch_dev(fitted, out_ts, out_val, n)
        assert produced == n and metrics["rows_created_by_macaque_v"] == n
    for series in (0, 1, streams // 2, 42_949, 42_950, streams - 1):   # (42 949.67 streams are 2^31 values)
        got = hip.download_array(out_val, per, np.float32, offset_elements=series * per)
        assert np.array_equal(got.view(np.uint32), datagen.bench_series(series, per).view(np.uint32)), series
        assert np.array_equal(hip.download_array(out_ts, 3, np.int64, offset_elements=series * per), [0, 1000, 2000])
    for pointer in (out_ts, out_val, offsets_dev):
        hip.dev_free(pointer)
    fitted.free()


def _short_simple_segments(rng, n, longest=20):
    """n PMC-Mean / Swing segments of 1..longest points with regular timestamps and no residuals, as rows."""
    lengths = rng.integers(1, longest + 1, n)
    types = rng.integers(0, 2, n)
    deltas = rng.integers(1, 2000, n)
    starts = np.cumsum(lengths * deltas + rng.integers(1, 50, n)) - lengths * deltas
    first = rng.uniform(-100.0, 100.0, n).astype(np.float32)
    last = (first + rng.uniform(0.5, 30.0, n).astype(np.float32)).astype(np.float32)
    decreasing = rng.random(n) < 0.5
    rows = []
    for k in range(n):
        length, start, delta = int(lengths[k]), int(starts[k]), int(deltas[k])
        end = start + (length - 1) * delta
        timestamps = b"" if length <= 2 else bytes([length])
        if length == 2 and rng.random() < 0.5:
            timestamps = bytes([2])   # (two points may also carry their length)
        if types[k] == 0 or length == 1:
            rows.append((0, start, end, timestamps, float(first[k]), float(first[k]), b"", b""))
        else:   # Swing: (min, max) are (first, last) or, with the one-byte flag, (last, first)
            rows.append((1, start, end, timestamps, float(first[k]), float(last[k]), bytes([0]) if decreasing[k] else b"", b""))
    return rows


@pytest.mark.parametrize("fused", ["auto", "forced", "off"])
def test_short_simple_segments_in_one_pass(hip, fused, monkeypatch, macaque_decoder):
    # k_grid_fused: a batch of short PMC-Mean / Swing segments with regular timestamps is reconstructed from the
    # raw rows in one pass (workgroups of 256 segments, their place in the output from a look-back over the
    # workgroups in front), the same points as the prepass / offsets / tiles pipeline writes.
    if fused == "forced":
        monkeypatch.setenv("MDB_GRID_FUSED", "1")
    elif fused == "off":
        monkeypatch.setenv("MDB_GRID_FUSED", "0")
    rng = np.random.default_rng(71)
    for n, longest in ((70_000, 20), (3, 5), (257, 3), (1000, 1), (5000, 64)):
        if fused == "auto" and n < 65536:
            continue
        rows = _short_simple_segments(rng, n, longest)
        batch = mdb.SegmentBatch.from_rows(rows)
        expected = ora.grid_batch(batch)
        hip.profile_enable(True)
        hip.profile_reset()
        got = hip.grid_batch(batch)
        kernels = hip.profile()
        hip.profile_enable(False)
        assert ("k_grid_fused" in kernels) == (fused != "off") and ("k_grid_tiles" in kernels) == (fused == "off")
        cases.assert_grid_equal(got, expected)
        assert np.array_equal(got[2], expected[2]) and got[3] == expected[3]
        resident = hip.upload_segments(batch)
        cases.assert_grid_equal(hip.grid_resident(resident), expected)
        resident.free()
        with pytest.raises(mdb.HipError, match="too small"):
            hip.grid_batch(batch, cap=len(expected[0]) - 1)
    # segments with serial work among them (a residual tail, a MacaqueV model): their model parts and timestamps
    # come from the same pass, their streams from the kernels behind it
    rows = _short_simple_segments(rng, 70_000 if fused == "auto" else 600, 12)
    at = rows[400][2] + 1
    tail = ora.try_compress_univariate_time_series(
        np.arange(30, dtype=np.int64) * 100 + at,
        np.concatenate([np.arange(20) * 3.0 + 1.0, rng.uniform(-1e3, 1e3, 10)]).astype(np.float32), cases.LOSSLESS)
    noise = ora.try_compress_univariate_time_series(np.arange(40, dtype=np.int64) * 100 + int(tail.end_time[-1]) + 1,
                                                    rng.uniform(-1e3, 1e3, 40).astype(np.float32), cases.LOSSLESS)
    assert set(noise.model_type_id.tolist()) == {2} and (tail.residuals.lengths() > 0).any()
    shift = int(noise.end_time[-1]) + 1
    with_streams = rows[:401] + tail.rows() + noise.rows() + [(r[0], r[1] + shift, r[2] + shift) + r[3:] for r in rows[401:]]
    batch = mdb.SegmentBatch.from_rows(with_streams)
    expected = ora.grid_batch(batch)
    for resident in (None, hip.upload_segments(batch)):
        hip.profile_enable(True)
        hip.profile_reset()
        got = hip.grid_batch(batch) if resident is None else hip.grid_resident(resident)
        kernels = hip.profile()
        hip.profile_enable(False)
        assert ("k_grid_fused" in kernels) == (fused != "off") and ("k_grid_tiles" in kernels) == (fused == "off")
        # (the MacaqueV model's 40 values: by pieces from a resident batch's cursors - or, in the one mode of this
        # file that asks for them from a single value on, from the cursors the call's host threads leave)
        by_pieces = resident is not None or macaque_decoder == "host-cursors"
        assert ("k_grid_mv_pieces" in kernels) == by_pieces and ("k_grid_serial" in kernels) == (resident is None)
        cases.assert_grid_equal(got, expected)
        if resident is not None:
            resident.free()
    # a delta-of-delta timestamp stream (three irregular points, inside its view): the general pipeline takes the batch
    irregular = ora.try_compress_univariate_time_series(np.array([0, 130, 200], dtype=np.int64) + shift + rows[-1][2] + 1,
                                                        np.array([1.0, 1.0, 1.0], dtype=np.float32), cases.LOSSLESS)
    assert not mdb.are_compressed_timestamps_regular(irregular.timestamps.value(0))
    batch = mdb.SegmentBatch.from_rows(with_streams + irregular.rows())
    hip.profile_enable(True)
    hip.profile_reset()
    got = hip.grid_batch(batch)
    kernels = hip.profile()
    hip.profile_enable(False)
    assert "k_grid_tiles" in kernels
    cases.assert_grid_equal(got, ora.grid_batch(batch))


def test_resident_batches_of_short_segments_with_residual_tails(hip):
    # The fitter's own segments of series that break into short models with residual tails of every length (a few values
    # inside the view, a couple of hundred behind it) and MacaqueV segments between them, as a resident batch: the tails
    # are pieces of the batch's cursor index like the MacaqueV streams. Whole, under a time range, and counted.
    rng = np.random.default_rng(113)

    def series(kinds, n_runs):
        parts = []
        for _ in range(n_runs):
            kind = kinds[int(rng.integers(0, len(kinds)))]
            level = float(rng.uniform(-50, 50))
            if kind == "short":     # a constant run, a few values of noise behind it
                parts += [np.full(int(rng.integers(9, 30)), level), rng.uniform(-1e3, 1e3, int(rng.integers(1, 9)))]
            elif kind == "line":    # a line, some values of noise
                parts += [level + 0.5 * np.arange(int(rng.integers(9, 40))), rng.uniform(-1e3, 1e3, int(rng.integers(1, 40)))]
            elif kind == "long":    # ... and tails of a couple of hundred values
                parts += [np.full(int(rng.integers(9, 20)), level), rng.uniform(-1e3, 1e3, int(rng.integers(150, 250)))]
            else:                   # noise alone: MacaqueV segments of their own
                parts += [rng.uniform(-1e3, 1e3, int(rng.integers(300, 900)))]
        return np.concatenate(parts).astype(np.float32)

    for kinds, n_runs in ((("short", "line"), 3000), (("short", "noise", "long", "line"), 1500), (("noise",), 300)):
        values = series(kinds, n_runs)
        timestamps = 1_700_000_000_000 + 100 * np.arange(len(values), dtype=np.int64)
        offsets = np.array(list(range(0, len(values), 50_000)) + [len(values)], dtype=np.uint64)
        for eb in (cases.LOSSLESS, cases.error_bounds()["rel1"]):
            segments = hip.compress_chunks(timestamps, values, offsets, eb)
            expected = ora.grid_batch(segments)
            resident = hip.upload_segments(segments)
            hip.profile_enable(True)
            hip.profile_reset()
            got = hip.grid_resident(resident)
            kernels = hip.profile()
            hip.profile_enable(False)
            assert "k_grid_mv_pieces" in kernels
            cases.assert_grid_equal(got, expected)
            cases.assert_grid_equal(hip.grid_resident(resident), expected)
            middle = (int(timestamps[len(values) // 3]), int(timestamps[2 * len(values) // 3]))
            keep = (expected[0] >= middle[0]) & (expected[0] <= middle[1])
            cases.assert_grid_equal(hip.grid_resident(resident, middle), (expected[0][keep], expected[1][keep]))
            _assert = hip.agg_batch_dev(resident, mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX)
            want = ora.agg_batch(segments, mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX)
            assert (_assert.count, _assert.min, _assert.max) == (want.count, want.min, want.max)
            resident.free()
