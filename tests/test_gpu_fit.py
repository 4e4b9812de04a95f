"""Parity of the HIP fitter (PMC-Mean / Swing / MacaqueV compression) with the CPU oracle through
the C ABI (mdb_compress_chunks / mdb_compress_series).

Bar: the same segments, byte for byte - model type ids, start/end times, min/max values and the
timestamps / values / residuals payloads - and the reference's own acceptance criterion
(compression.rs:865-929): decoded timestamps equal, every value within the error bound."""

import numpy as np
import pytest

import cases
import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[("auto", None, None), ("1", "off", None), ("64", "8", None),
                                      ("1000", None, "lean-off"), ("1", None, "plain"), ("auto", None, "wave"),
                                      ("64", None, "wave-or-pieces")],
                ids=["split-auto", "lane-per-chunk+lane-per-gap", "pieces-of-64+gaps-from-8",
                     "pieces-of-1000+k_fit_models-fast", "lane-per-chunk+k_fit_models-plain", "wave-per-chunk",
                     "wave-leaving-short-models-to-pieces-of-64"])
def fit_mode(request, monkeypatch):
    """Every fit test runs with the library choosing between one lane per chunk and split mode
    (speculative pieces + chain walk, mdb_fit.hip), with split mode off, and with it forced; with
    long lossless MacaqueV-only segments encoded by one wave each (k_fit_gap) from the default length,
    never, and from 8 values; and with each of the forms of the greedy loop (k_fit_models_lean where it
    applies, k_fit_models with the fast forms of the fitters, k_fit_models plain, and k_fit_models_wave:
    one wave per chunk, 64 consecutive points per step - to the end of every chunk, and leaving chunks
    with short models to split mode)."""
    pieces, gaps, loop = request.param
    for name in ("MDB_FIT_LEAN", "MDB_FIT_FAST", "MDB_FIT_WAVE", "MDB_FIT_WAVE_WINDOW_POINTS",
                 "MDB_FIT_WAVE_POINTS_PER_STEP"):
        monkeypatch.delenv(name, raising=False)
    if loop == "wave":
        monkeypatch.setenv("MDB_FIT_WAVE", "1")  # one wave per chunk (k_fit_models_wave) wherever it applies
    if loop == "wave-or-pieces":
        # ... which looks at its progress every 64 points and leaves a chunk with fewer than 40 points per step
        # to split mode: calls with several chunks end up with some chunks from either.
        monkeypatch.setenv("MDB_FIT_WAVE", "2")
        monkeypatch.setenv("MDB_FIT_WAVE_WINDOW_POINTS", "64")
        monkeypatch.setenv("MDB_FIT_WAVE_POINTS_PER_STEP", "40")
    if loop in ("lean-off", "plain"):
        monkeypatch.setenv("MDB_FIT_LEAN", "0")
    if loop == "plain":
        monkeypatch.setenv("MDB_FIT_FAST", "0")

    if pieces == "auto":
        monkeypatch.delenv("MDB_FIT_PIECE_POINTS", raising=False)
    else:
        monkeypatch.setenv("MDB_FIT_PIECE_POINTS", pieces)
    if gaps is None:
        monkeypatch.delenv("MDB_FIT_GAP_MIN_VALUES", raising=False)
    else:
        monkeypatch.setenv("MDB_FIT_GAP_MIN_VALUES", gaps)
    return pieces


def assert_same_segments(got, expected):
    assert len(got) == len(expected)
    assert np.array_equal(got.model_type_id, expected.model_type_id)
    assert np.array_equal(got.start_time, expected.start_time)
    assert np.array_equal(got.end_time, expected.end_time)
    assert np.array_equal(got.min_value.view(np.uint32), expected.min_value.view(np.uint32))
    assert np.array_equal(got.max_value.view(np.uint32), expected.max_value.view(np.uint32))
    assert got.timestamps.to_bytes_list() == expected.timestamps.to_bytes_list()
    assert got.values.to_bytes_list() == expected.values.to_bytes_list()
    assert got.residuals.to_bytes_list() == expected.residuals.to_bytes_list()
    assert np.isnan(got.error).all()
    if expected.chunk_index is not None:
        assert np.array_equal(got.chunk_index, expected.chunk_index)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("noise", [None, (1.0, 1.05)])
@pytest.mark.parametrize("eb_name", ["lossless", "abs5", "rel5", "rel1", "abs0.01"])
def test_fit_matches_oracle_on_synthetic_series(hip, eb_name, noise, irregular):
    # The recipe of compression.rs:733-863.
    eb = cases.error_bounds()[eb_name]
    timestamps, values = cases.synthetic_series(50_000, irregular, noise, seed=31)
    expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
    got = hip.try_compress_univariate_time_series(timestamps, values, eb)
    assert_same_segments(got, expected)
    ts, reconstructed, _, _ = hip.grid_batch(got)
    assert np.array_equal(ts, timestamps)


def test_fit_edge_cases(hip):
    for eb_name in ("lossless", "rel5", "abs5"):
        eb = cases.error_bounds()[eb_name]
        for name, ts, values in cases.edge_case_series():
            expected = ora.try_compress_univariate_time_series(ts, values, eb)
            got = hip.try_compress_univariate_time_series(ts, values, eb)
            assert_same_segments(got, expected)


def test_fit_empty_and_mismatched_input(hip):
    assert len(hip.try_compress_univariate_time_series([], [], cases.LOSSLESS)) == 0  # :422-434
    with pytest.raises(mdb.HipError, match="different lengths"):                      # :202-206
        hip.try_compress_univariate_time_series([1, 2], [1.0], cases.LOSSLESS)
    with pytest.raises(mdb.HipError, match="error bound"):
        hip.try_compress_univariate_time_series([1, 2], [1.0, 2.0], mdb._abi.ErrorBoundC(1, -1.0))


def test_fit_known_answer_segment(hip):  # compression.rs:932-978
    batch = hip.try_compress_univariate_time_series([100, 200, 300, 400, 500],
                                                    [73.0, 37.0, 37.0, 37.0, 73.0], cases.LOSSLESS)
    assert batch.rows() == [(2, 100, 500, bytes([5]), 37.0, 73.0, bytes.fromhex("42920000d03c3a43"), b"")]


def test_fit_many_chunks_in_one_launch(hip):
    # The server hands the compressor one <= 65 536 point buffer per series (storage/mod.rs:58);
    # the batch entry point takes many at once. Chunks of ragged lengths, including empty ones.
    eb = cases.error_bounds()["rel1"]
    rng = np.random.default_rng(41)
    lengths = [0, 1, 2, 7, 8, 9, 300, 65536, 1000, 0, 4097] + [int(x) for x in rng.integers(1, 5000, 40)]
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    total = int(offsets[-1])
    values = np.concatenate([datagen.sine_series(s, n)[1] if n else np.zeros(0, np.float32)
                             for s, n in enumerate(lengths)])
    timestamps = np.concatenate([np.arange(n, dtype=np.int64) * 1000 for n in lengths])
    assert len(values) == total
    expected = ora.compress_chunks(timestamps, values, offsets, eb)
    got = hip.compress_chunks(timestamps, values, offsets, eb)
    assert_same_segments(got, expected)


def test_fit_sine_workload_full_chunks(hip):
    # The benchmark's series cut into 65 536 point chunks.
    eb = cases.error_bounds()["rel1"]
    n_series, n_points = 8, 300_000
    timestamps = np.tile(np.arange(n_points, dtype=np.int64) * 1000, n_series)
    values = np.concatenate([datagen.sine_series(s, n_points)[1] for s in range(n_series)])
    offsets = []
    for s in range(n_series):
        offsets += [s * n_points + c for c in range(0, n_points, 65536)]
    offsets = np.array(offsets + [n_series * n_points], dtype=np.uint64)
    expected = ora.compress_chunks(timestamps, values, offsets, eb, n_threads=8)
    got = hip.compress_chunks(timestamps, values, offsets, eb)
    assert_same_segments(got, expected)
    ts, reconstructed, rows, _ = hip.grid_batch(got)
    assert np.array_equal(ts, timestamps)
    relative = np.abs((values - reconstructed) / values) * np.float32(100.0)
    assert (relative <= np.float32(1.0)).all()


def test_fit_device_resident_with_synthesised_timestamps(hip):
    # mdb_compress_chunks_dev with ts == NULL must equal materialised regular timestamps.
    eb = cases.error_bounds()["rel1"]
    n_series, n_points, chunk = 3, 200_000, 65536
    values = np.concatenate([datagen.sine_series(s, n_points)[1] for s in range(n_series)])
    offsets, first_index = [], []
    for s in range(n_series):
        for c in range(0, n_points, chunk):
            offsets.append(s * n_points + c)
            first_index.append(c)
    offsets = np.array(offsets + [n_series * n_points], dtype=np.uint64)
    first_index = np.array(first_index, dtype=np.uint64)
    timestamps = np.tile(np.arange(n_points, dtype=np.int64) * 1000, n_series)
    expected = ora.compress_chunks(timestamps, values, offsets, eb, n_threads=8)
    values_dev = hip.upload_array(values)
    offsets_dev = hip.upload_array(offsets)
    first_dev = hip.upload_array(first_index)
    dev = hip.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, eb, 0, 1000, first_dev)
    got = dev.download()
    assert_same_segments(got, expected)
    # and grid straight from the device-resident segments
    total = hip.grid_count_dev(dev)
    assert total == n_series * n_points
    out_ts, out_val = hip.dev_alloc(8 * total), hip.dev_alloc(4 * total)
    hip.grid_batch_dev(dev, out_ts, out_val, total)
    assert np.array_equal(hip.download_array(out_ts, total, np.int64), timestamps)
    for pointer in (values_dev, offsets_dev, first_dev, out_ts, out_val):
        hip.dev_free(pointer)
    dev.free()


def test_synthetic_generator_statistics(hip):
    n_series, n_points = 4, 100_000
    pointer = hip.dev_alloc(4 * n_series * n_points)
    hip.synth_values_dev(pointer, 0, n_series, n_points)
    values = hip.download_array(pointer, n_series * n_points, np.float32).reshape(n_series, n_points)
    hip.dev_free(pointer)
    assert np.isfinite(values).all()
    assert 89.9 <= values.min() <= 90.2 and 109.8 <= values.max() <= 110.1
    i = np.arange(n_points, dtype=np.float64)
    for s in range(n_series):
        clean = 100.0 + 10.0 * np.sin(2 * np.pi * i / (2000.0 + 37.0 * (s % 64)) + 2 * np.pi * ((s * 0.61803) % 1.0))
        assert np.abs(values[s] - clean).max() <= 0.0501


def test_synthetic_generator_is_the_host_definition_bit_for_bit(hip):
    """The bench fits what mdb_synth_values_dev writes; tests/datagen.bench_series is the same
    recipe on the host (fixed polynomial sine, splitmix64 noise, IEEE +, *, /, floor only), so the
    oracle can be run on exactly the bytes the bench fits."""
    n_points = 300_001  # not a multiple of 4: the kernel's scalar tail too
    for first_series, n_series in ((0, 3), (63, 2), (999, 1), (12_499, 1)):
        pointer = hip.dev_alloc(4 * n_series * n_points + 16)
        hip.synth_values_dev(pointer, first_series, n_series, n_points)
        got = hip.download_array(pointer, n_series * n_points, np.float32).reshape(n_series, n_points)
        hip.dev_free(pointer)
        for s in range(n_series):
            expected = datagen.bench_series(first_series + s, n_points)
            assert np.array_equal(got[s].view(np.uint32), expected.view(np.uint32)), (first_series + s)


def test_bench_workload_fit_and_grid_match_the_oracle_on_the_generated_bytes(hip):
    """Two series of the benchmark workload exactly as bench.py builds them (device generator,
    65 536-point chunks, regular timestamps synthesised by the fitter): the oracle fits the host
    definition of the same series into the same segments, and grids them to the same points."""
    n_series, n_points, chunk = 2, 400_000, 65536
    eb = mdb.error_bound("relative", 1.0)
    values_dev = hip.dev_alloc(4 * n_series * n_points)
    hip.synth_values_dev(values_dev, 5, n_series, n_points)
    starts = np.arange(0, n_points, chunk, dtype=np.uint64)
    offsets = np.concatenate([(np.arange(n_series, dtype=np.uint64)[:, None] * np.uint64(n_points) + starts).reshape(-1),
                              np.array([n_series * n_points], dtype=np.uint64)])
    first_index = np.tile(starts, n_series)
    offsets_dev, first_dev = hip.upload_array(offsets), hip.upload_array(first_index)
    dev = hip.compress_chunks_dev(0, values_dev, offsets_dev, len(offsets) - 1, eb, 0, 1000, first_dev)
    host_values = np.concatenate([datagen.bench_series(5 + s, n_points) for s in range(n_series)])
    host_ts = np.tile(np.arange(n_points, dtype=np.int64) * 1000, n_series)
    expected = ora.compress_chunks(host_ts, host_values, offsets, eb)
    got = dev.download()
    assert_same_segments(got, expected)
    total = hip.grid_count_dev(dev)
    out_ts, out_val = hip.dev_alloc(8 * total), hip.dev_alloc(4 * total)
    hip.grid_batch_dev(dev, out_ts, out_val, total)
    oracle_ts, oracle_values = ora.grid_batch(expected)[:2]
    assert np.array_equal(hip.download_array(out_ts, total, np.int64), oracle_ts)
    assert np.array_equal(hip.download_array(out_val, total, np.float32).view(np.uint32),
                          oracle_values.view(np.uint32))
    for pointer in (values_dev, offsets_dev, first_dev, out_ts, out_val):
        hip.dev_free(pointer)
    dev.free()


def test_split_and_compress_univariate_time_series(hip):
    """try_split_and_compress_univariate_time_series (compression.rs:147-179): several field columns
    of one series, one error bound per field, the timestamps shared."""
    for irregular in (False, True):
        timestamps, first = cases.synthetic_series(20_000, irregular, (1.0, 1.05), seed=211)
        _, second = cases.synthetic_series(20_000, irregular, None, seed=212)
        third = datagen.sine_series(3, 20_000)[1]
        bounds = [cases.error_bounds()[name] for name in ("rel5", "lossless", "abs0.01")]
        fields = [first, second, third]
        got = hip.try_split_and_compress_univariate_time_series(timestamps, fields, bounds)
        assert len(got) == 3
        for batch, values, eb in zip(got, fields, bounds):
            expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
            expected.chunk_index = None
            assert_same_segments(batch, expected)
    assert hip.try_split_and_compress_univariate_time_series(np.arange(5) * 100, [], []) == []
    empty = hip.try_split_and_compress_univariate_time_series([], [[]], [cases.LOSSLESS])
    assert len(empty) == 1 and len(empty[0]) == 0
    with pytest.raises(mdb.HipError, match="different lengths"):
        hip.try_split_and_compress_univariate_time_series([1, 2, 3], [[1.0, 2.0]], [cases.LOSSLESS])


def test_payload_columns_roll_over_into_several_data_buffers(hip, monkeypatch):
    """A BinaryView column whose payloads exceed what one data buffer may hold (2 GiB for arrow; the
    library begins a new buffer every GiB) comes back with several buffers, as arrow's builders make
    them (types.rs:444-516). Forced here with a buffer size of a few hundred bytes: same rows as the
    oracle's, several buffers per column, on the device and after the download, and grid() reads them."""
    rng = np.random.default_rng(17)
    n_chunks, n_points = 6, 4000
    timestamps = np.concatenate([np.cumsum(rng.integers(100, 900, n_points)).astype(np.int64) for _ in range(n_chunks)])
    values = rng.uniform(-1e6, 1e6, n_chunks * n_points).astype(np.float32)
    values[: n_points] = np.repeat(rng.uniform(1, 2, n_points // 400), 400)[:n_points]  # models with residual tails
    offsets = np.arange(0, n_chunks * n_points + 1, n_points, dtype=np.uint64)
    for eb in (cases.LOSSLESS, cases.error_bounds()["rel1"]):
        expected = ora.compress_chunks(timestamps, values, offsets, eb)
        for buffer_bytes in ("300", "5000", None):
            if buffer_bytes is None:
                monkeypatch.delenv("MDB_FIT_DATA_BUFFER_BYTES", raising=False)
            else:
                monkeypatch.setenv("MDB_FIT_DATA_BUFFER_BYTES", buffer_bytes)
            ts_dev, values_dev, offsets_dev = (hip.upload_array(a) for a in (timestamps, values, offsets))
            dev = hip.compress_chunks_dev(ts_dev, values_dev, offsets_dev, n_chunks, eb)
            if buffer_bytes is not None:
                assert dev.seg.timestamps.n_buffers > 3 and dev.seg.values.n_buffers > 3
                limit = int(buffer_bytes)
                longest = max(len(p) for p in expected.timestamps.to_bytes_list())
                assert all(dev.seg.timestamps.buffer_sizes[k] <= limit + longest for k in range(dev.seg.timestamps.n_buffers))
            else:
                assert dev.seg.timestamps.n_buffers == 1
            hip.validate_segments_dev(dev)
            got = dev.download()
            assert_same_segments(got, expected)
            assert len(got.timestamps.buffers) == 1  # merged on the way down while it all fits into one
            if buffer_bytes is not None:
                monkeypatch.setenv("MDB_SEGMENTS_MERGE_LIMIT", "1000")  # as if it did not
                apart = dev.download()
                monkeypatch.delenv("MDB_SEGMENTS_MERGE_LIMIT")
                assert len(apart.timestamps.buffers) == dev.seg.timestamps.n_buffers > 3
                assert_same_segments(apart, expected)
                cases.assert_grid_equal(hip.grid_batch(apart), ora.grid_batch(expected))
            total = hip.grid_count_dev(dev)
            out_ts, out_val = hip.dev_alloc(8 * total), hip.dev_alloc(4 * total)
            hip.grid_batch_dev(dev, out_ts, out_val, total)
            assert np.array_equal(hip.download_array(out_ts, total, np.int64), timestamps)
            oracle_values = ora.grid_batch(expected)[1]
            assert np.array_equal(hip.download_array(out_val, total, np.float32).view(np.uint32), oracle_values.view(np.uint32))
            for pointer in (ts_dev, values_dev, offsets_dev, out_ts, out_val):
                hip.dev_free(pointer)
            dev.free()
            # and through the host entry point (upload, fit, download in one call)
            assert_same_segments(hip.compress_chunks(timestamps, values, offsets, eb), expected)


def test_fit_pmc_chosen_while_swing_ran_far_ahead(hip):
    # PMC-Mean wins ties (types.rs:84-101) even when Swing accepted up to ~3 % more points, so the
    # next model starts well behind the last point that was fed. Long near-constant runs with a
    # small drift provoke exactly that; the fitter's prefetch ring must rewind correctly.
    rng = np.random.default_rng(91)
    parts = []
    for k in range(40):
        length = int(rng.integers(300, 3000))
        level = float(rng.uniform(50, 150))
        drift = float(rng.uniform(-0.003, 0.003))
        parts.append((level * (1.0 + drift * np.arange(length) / length)).astype(np.float32))
    values = np.concatenate(parts)
    timestamps = np.arange(len(values), dtype=np.int64) * 1000
    for eb_name in ("rel1", "rel5", "abs5"):
        eb = cases.error_bounds()[eb_name]
        expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
        got = hip.try_compress_univariate_time_series(timestamps, values, eb)
        assert_same_segments(got, expected)
    # the scenario really occurs: some PMC-Mean segment is followed by a model starting before
    # the point where Swing would have failed
    assert (expected.model_type_id == 0).any()


def test_fit_split_mode_on_chains_that_converge_late_or_never(hip):
    # Split mode relies on greedy chains from different start points meeting. Worst cases: one model
    # spans the whole chunk (no piece ever meets another before the end), every point is rejected
    # (every point is on every chain), and long models with a noisy stretch in the middle.
    rng = np.random.default_rng(77)
    n = 40_000
    timestamps = np.arange(n, dtype=np.int64) * 1000 + 1_000_000
    noisy = rng.normal(0.0, 1000.0, n).astype(np.float32)
    ramp = (np.arange(n, dtype=np.float64) * 0.25 + 10.0).astype(np.float32)
    mixed = np.full(n, 42.0, dtype=np.float32)
    mixed[15_000:15_700] = noisy[:700]
    mixed[30_000:] = ramp[30_000:]
    for name, values in (("constant", np.full(n, 42.0, dtype=np.float32)), ("ramp", ramp), ("noise", noisy),
                         ("mixed", mixed)):
        for eb_name in ("lossless", "rel1"):
            eb = cases.error_bounds()[eb_name]
            expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
            got = hip.try_compress_univariate_time_series(timestamps, values, eb)
            assert_same_segments(got, expected)


def test_fit_one_long_series_in_one_call(hip, fit_mode):
    # The embedded API compresses a whole series in one call (data_folder.rs:214): BASELINE config 1
    # is 1 series x 1 M points, lossless; here also with the benchmark's 1 % bound.
    if fit_mode == "64":
        pytest.skip("16 384 pieces of 64 points add nothing over the other modes here")
    timestamps, values = datagen.sine_series(3, 1_000_000)
    for eb_name in ("rel1", "lossless"):
        eb = cases.error_bounds()[eb_name]
        expected = ora.try_compress_univariate_time_series(timestamps, values, eb)
        got = hip.try_compress_univariate_time_series(timestamps, values, eb)
        assert_same_segments(got, expected)


def test_fit_regular_and_irregular_chunks_in_one_call(hip):
    # k_fit_regular decides per chunk whether its timestamps are equally spaced; only when ALL chunks
    # are does the library stop loading timestamps. Mix both kinds (and chunks of 0, 1 and 2 points,
    # and one whose only irregularity is its very last gap) in one call.
    eb = cases.error_bounds()["rel1"]
    rng = np.random.default_rng(53)
    lengths = [5000, 3000, 0, 1, 2, 4000, 70_000, 2500, 6000]
    kinds = ["regular", "irregular", "regular", "regular", "irregular", "last-gap", "regular", "irregular", "regular"]
    timestamps, values = [], []
    for s, (n, kind) in enumerate(zip(lengths, kinds)):
        if kind == "regular":
            ts = 1_000_000 * s + np.arange(n, dtype=np.int64) * (1000 + s)
        elif kind == "irregular":
            ts = 1_000_000 * s + np.cumsum(rng.integers(1, 2000, n)).astype(np.int64)
        else:
            ts = 1_000_000 * s + np.arange(n, dtype=np.int64) * 1000
            ts[-1] += 7
        timestamps.append(ts)
        values.append(datagen.sine_series(s, n)[1] if n else np.zeros(0, np.float32))
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    timestamps, values = np.concatenate(timestamps), np.concatenate(values)
    expected = ora.compress_chunks(timestamps, values, offsets, eb)
    got = hip.compress_chunks(timestamps, values, offsets, eb)
    assert_same_segments(got, expected)
    # and all-regular with a different interval per chunk (the path that stops loading timestamps)
    regular = [1_000_000 * s + np.arange(n, dtype=np.int64) * (1000 + 13 * s) for s, n in enumerate(lengths)]
    regular = np.concatenate(regular)
    assert_same_segments(hip.compress_chunks(regular, values, offsets, eb),
                         ora.compress_chunks(regular, values, offsets, eb))


@pytest.mark.parametrize("origin", [(1 << 52) - 3_000_000, -(1 << 52) - 40_000, 1 << 60, -(1 << 62)])
def test_fit_irregular_timestamps_around_and_beyond_the_exact_f64_range(hip, origin):
    # The straight-line fitter computes with (f64)timestamp, which is the timestamp itself only up to 2^52 in
    # magnitude (k_fit_regular checks every one); a series that crosses that line, or lies far beyond it,
    # must take the general kernel and still produce the oracle's segments. Chunks of odd lengths and
    # starts, so that the timestamp ring's first and last groups are partial.
    rng = np.random.default_rng(91)
    lengths = [7001, 1, 2, 3, 4999, 13, 20_000]
    n = sum(lengths)
    timestamps = origin + np.cumsum(rng.integers(1, 3000, n)).astype(np.int64)
    values = (10 + np.sin(np.arange(n) / 400.0) + rng.normal(0, 0.01, n)).astype(np.float32)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    for eb_name in ("rel1", "abs0.01"):
        eb = cases.error_bounds()[eb_name]
        assert_same_segments(hip.compress_chunks(timestamps, values, offsets, eb),
                             ora.compress_chunks(timestamps, values, offsets, eb))


def test_fit_chunk_list_gathers_chunks_where_they_lie(hip):
    # mdb_compress_chunk_list (what the patched try_compress_multivariate_time_series and
    # process_compressor_messages call): every chunk its own pair of arrays, fields of one series sharing their
    # timestamp array, regular and irregular, empty ones, more than one gather slice.
    eb = cases.error_bounds()["rel1"]
    rng = np.random.default_rng(59)
    chunks = []
    for s, n in enumerate([5000, 0, 1, 2, 70_000, 2500, 300_000, 9, 65_536]):
        ts = 1_000_000 * s + np.arange(n, dtype=np.int64) * (1000 + s)
        for field in range(3):                       # three fields share ts
            values = datagen.sine_series(7 * s + field, n)[1] if n else np.zeros(0, np.float32)
            chunks.append((ts, values))
    flat_ts = np.concatenate([ts for ts, _ in chunks])
    flat_values = np.concatenate([v for _, v in chunks])
    offsets = np.concatenate([[0], np.cumsum([len(v) for _, v in chunks])]).astype(np.uint64)
    assert_same_segments(hip.compress_chunk_list(chunks, eb), ora.compress_chunks(flat_ts, flat_values, offsets, eb))
    # one chunk with irregular timestamps among them: the timestamps cross as well
    irregular = 5_000_000 + np.cumsum(rng.integers(1, 2000, 40_000)).astype(np.int64)
    chunks.insert(4, (irregular, datagen.sine_series(99, len(irregular))[1]))
    flat_ts = np.concatenate([ts for ts, _ in chunks])
    flat_values = np.concatenate([v for _, v in chunks])
    offsets = np.concatenate([[0], np.cumsum([len(v) for _, v in chunks])]).astype(np.uint64)
    for bound in (eb, cases.LOSSLESS):
        assert_same_segments(hip.compress_chunk_list(chunks, bound),
                             ora.compress_chunks(flat_ts, flat_values, offsets, bound))
    assert len(hip.compress_chunk_list([], eb)) == 0
    with pytest.raises(mdb.HipError):
        hip.compress_chunk_list([(np.arange(3), np.zeros(2, np.float32))], eb)


def test_fit_chunk_list_of_many_chunks_and_its_host_phases(hip, monkeypatch):
    """More chunks than the few-chunks path takes, at odd addresses and lengths (the gather copies with 16-byte
    streaming stores from wherever the chunks lie: heads and tails that are not 16 bytes), in slices; with the profile
    on, the call's host-side phases are on record under "host:" names (include/mdb.h, mdb_profile_get)."""
    monkeypatch.setenv("MDB_FIT_SMALL", "4")
    eb = cases.error_bounds()["rel1"]
    rng = np.random.default_rng(77)
    backing = np.concatenate([datagen.sine_series(s, 40_000)[1] for s in range(12)])
    chunks, at = [], 0
    for k in range(150):
        n = int(rng.integers(1, 6000))
        at += int(rng.integers(0, 5))                      # (chunks begin at any multiple of four bytes)
        if at + n > len(backing):
            at = int(rng.integers(0, 7))
        chunks.append((np.arange(n, dtype=np.int64) * 100 + k, backing[at:at + n]))
        at += n
    flat_ts = np.concatenate([ts for ts, _ in chunks])
    flat_values = np.concatenate([v for _, v in chunks])
    offsets = np.concatenate([[0], np.cumsum([len(v) for _, v in chunks])]).astype(np.uint64)
    hip.profile_enable(True)
    hip.profile_reset()
    got = hip.compress_chunk_list(chunks, eb)
    recorded = hip.profile()
    hip.profile_enable(False)
    hip.profile_reset()
    assert_same_segments(got, ora.compress_chunks(flat_ts, flat_values, offsets, eb))
    for phase in ("gather", "upload_tail", "fit", "download"):
        calls, ms = recorded["host:chunk_list_" + phase]
        assert calls == 1 and ms >= 0.0
