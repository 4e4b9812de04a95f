"""The C++ host operators (GridExec / GridStream, Model*Accumulators, try_compress_*) end to end on
the GPU, written after the reference's own tests of the same operators:
crates/modelardb_embedded/src/operations/data_folder.rs:1087-1234 (3-point series, aggregates),
crates/modelardb_server/tests/integration_test.rs:1128-1246 (segment vs grid aggregates),
crates/modelardb_compression/src/compression.rs:422-434, 932-978 (compress)."""

import os

import numpy as np
import pyarrow as pa
import pytest

import cases
import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import host

# tests/test_host_sanitizers_cpu.py runs this very file without a GPU against the host library built
# with the CPU sanitizers and the canned-answer stand-in for libmdb_hip (tests/stub): the grid-heavy
# cases shrink tenfold there so that the fixture file of canned answers stays small.
UNDER_STUB = bool(os.environ.get("MDB_HOST_LIBRARY_UNDER_TEST"))
pytestmark = [] if UNDER_STUB else pytest.mark.gpu


def _n(points):
    return points // 10 if UNDER_STUB else points


@pytest.fixture(autouse=True, params=[None, "0"], ids=["one-batch-ahead", "grid-when-polled"])
def grid_prefetch(request, monkeypatch):
    """Every test runs with GridStream keeping one batch of segments ahead on a second context (the
    copy of one batch overlapping the kernels of the next) and with it gridding a batch when it is
    polled for, as the reference does (grid_exec.rs:402-412): the same rows in the same order."""
    if request.param is None:
        monkeypatch.delenv("MDB_HOST_GRID_PREFETCH", raising=False)
    else:
        monkeypatch.setenv("MDB_HOST_GRID_PREFETCH", request.param)
    return request.param


def _series(seed, length=None, irregular=False):
    length = _n(30_000) if length is None else length
    eb = cases.error_bounds()["rel5"]
    timestamps, values = cases.synthetic_series(length, irregular, (1.0, 1.05), seed)
    return timestamps, values, ora.try_compress_univariate_time_series(timestamps, values, eb)


def _segment_batches(batch, tags, rows_per_batch):
    arrow = host.segments_with_tags(batch.to_arrow(), tags)
    return [arrow.slice(start, rows_per_batch) for start in range(0, arrow.num_rows, rows_per_batch)]


def _concat(batches):
    table = pa.Table.from_batches(batches)
    return (table.column("timestamp").cast(pa.int64()).to_numpy(),
            table.column("value").to_numpy(), table)


@pytest.mark.parametrize("batch_size", [8192, 1000])
def test_grid_stream_reconstructs_two_series_with_tags(hip, batch_size):
    stream = host.GridStream(hip, tag_names=("tag",), batch_size=batch_size)
    expected_ts, expected_values, expected_tags = [], [], []
    total_segments = 0
    for seed, tag in ((61, "A"), (62, "a-long-tag-value-beyond-12-bytes")):
        _, _, batch = _series(seed)
        total_segments += len(batch)
        for part in _segment_batches(batch, {"tag": tag}, 64):
            stream.push(part)
        ts, values, _, _ = ora.grid_batch(batch)
        expected_ts.append(ts)
        expected_values.append(values)
        expected_tags += [tag] * len(ts)
    stream.finish_input()
    batches, state = stream.collect()
    assert state == host.GridStream.READY_NONE
    assert all(b.num_rows <= batch_size for b in batches)
    # (one input batch per poll, grid_exec.rs:407-417: full batches only while an input batch holds enough points)
    assert UNDER_STUB or all(b.num_rows == batch_size for b in batches[:-1])
    ts, values, table = _concat(batches)
    assert table.schema.names == ["timestamp", "value", "tag"]
    assert np.array_equal(ts, np.concatenate(expected_ts))
    assert np.array_equal(values.view(np.uint32), np.concatenate(expected_values).view(np.uint32))
    assert table.column("tag").to_pylist() == expected_tags
    metrics = stream.metrics()
    assert metrics["rows_created"] == metrics["output_rows"] == len(ts)
    assert metrics["regular_segments"] + metrics["irregular_segments"] == total_segments
    assert metrics["elapsed_compute_ns"] > 0


@pytest.mark.skipif(UNDER_STUB, reason="the stand-in's canned answers are for the default call sequence")
@pytest.mark.parametrize("contexts, ahead", [("1", "2"), ("3", "2"), ("4", "3"), ("2", "8")])
def test_more_submits_ahead_on_more_contexts_return_the_same_rows(contexts, ahead, monkeypatch):
    """MDB_GRID_PIPELINE_CONTEXTS (1..4: the contexts mdb_grid_submit's workers run jobs on) and MDB_HOST_GRID_PREFETCH
    (the submits GridStream keeps ahead of the one it waits for): whatever they are set to the stream returns the
    same rows in the same order, and the launches of every worker's context are in the profile of the context the
    submits were made on - one `host:` entry per submit and phase."""
    import modelardb_rs_amd as mdb
    monkeypatch.setenv("MDB_GRID_PIPELINE_CONTEXTS", contexts)
    monkeypatch.setenv("MDB_HOST_GRID_PREFETCH", ahead)
    monkeypatch.setenv("MDB_HOST_GRID_COALESCE_SEGMENTS", "2")
    context = mdb.Context(0)  # (a context of its own: the workers are made by its first submit)
    try:
        context.profile_enable(True)
        stream = host.GridStream(context, tag_names=("sensor",), batch_size=1000)
        expected_ts, expected_values, expected_tags, submits = [], [], [], 0
        for seed in range(3):
            _, _, batch = _series(900 + seed, length=20_000, irregular=seed == 1)
            for part in _segment_batches(batch, {"sensor": f"sensor-number-{seed}-of-three"}, 2):
                stream.push(part)
                submits += 1
            ts, values, _, _ = ora.grid_batch(batch)
            expected_ts.append(ts)
            expected_values.append(values)
            expected_tags += [f"sensor-number-{seed}-of-three"] * len(ts)
        stream.finish_input()
        batches, state = stream.collect()
        assert state == host.GridStream.READY_NONE
        ts, values, table = _concat(batches)
        assert np.array_equal(ts, np.concatenate(expected_ts))
        assert np.array_equal(values.view(np.uint32), np.concatenate(expected_values).view(np.uint32))
        assert table.column("sensor").to_pylist() == expected_tags
        stream.close()
        profile = context.profile()
        assert submits > 8
        made = profile["host:grid_launches"][0]  # (a batch of one row is gathered with the one behind it)
        assert submits // 2 <= made <= submits, profile
        assert profile["host:grid_kernels_and_copy_down"][0] == made, profile
        assert max(calls for name, (calls, _) in profile.items() if name.startswith("k_grid")) >= made, profile
        context.profile_reset()
        assert context.profile() == {}
    finally:
        context.close()


@pytest.mark.parametrize("coalesce", ["learned", "3", "1000000"])
def test_grid_stream_gathers_input_batches_and_keeps_one_submit_ahead(hip, coalesce, monkeypatch, grid_prefetch):
    """What the patched GridStream::poll_next (rust/patches/0001-grid_exec.patch) does, call for call: input
    batches that are ready are gathered into ONE mdb_grid_submit (several RecordBatches per launch, SURVEY 8(f)
    N2), the next submit is made before the current one is waited for, and the tag views of every gathered batch
    are repeated per row with their strings left in that batch's own data buffers. The rows are the same in the
    same order however the batches are gathered."""
    if coalesce != "learned":
        monkeypatch.setenv("MDB_HOST_GRID_COALESCE_SEGMENTS", coalesce)
    log = os.environ.get("MDB_STUB_CALL_LOG")
    if log and os.path.exists(log):
        os.remove(log)
    stream = host.GridStream(hip, tag_names=("site", "sensor"), batch_size=1000)
    expected_ts, expected_values, expected_tags = [], [], []
    pushed = []
    for seed in range(4):
        _, _, batch = _series(200 + seed, length=_n(12_000), irregular=seed % 2 == 1)
        tags = {"site": f"a-site-name-that-is-not-inlined-{seed}", "sensor": f"s{seed}"}
        for part in _segment_batches(batch, tags, 2):
            stream.push(part)
            pushed.append(part.num_rows)
        ts, values, _, _ = ora.grid_batch(batch)
        expected_ts.append(ts)
        expected_values.append(values)
        expected_tags += [(tags["site"], tags["sensor"])] * len(ts)
    stream.finish_input()
    batches, state = stream.collect()
    assert state == host.GridStream.READY_NONE
    ts, values, table = _concat(batches)
    assert np.array_equal(ts, np.concatenate(expected_ts))
    assert np.array_equal(values.view(np.uint32), np.concatenate(expected_values).view(np.uint32))
    assert list(zip(table.column("site").to_pylist(), table.column("sensor").to_pylist())) == expected_tags
    assert stream.metrics()["rows_created"] == len(ts)
    if log:  # (tests/test_shim_call_sequence_cpu.py: the launches the kernels' side saw, in order)
        calls = [line.split() for line in open(log).read().splitlines()]
        launches = [(int(n_inputs), int(segments)) for what, n_inputs, segments, _ in calls if what == "grid"]
        assert sum(segments for _, segments in launches) == sum(pushed)
        assert all(int(reserve_front) >= 1000 for what, _, _, reserve_front in calls if what == "grid")
        if coalesce == "3":
            assert all(n_inputs <= 2 for n_inputs, _ in launches)           # batches of 1-2 segments until 3 are there
            assert len(launches) >= len(pushed) / 2
        elif coalesce == "1000000":
            assert launches == [(len(pushed), sum(pushed))]                  # everything that was ready: one launch
        else:
            # the first submit is one batch (nothing is known about the data yet), with one submit kept ahead so
            # is the second; then the stream has seen what a segment decompresses to and takes all that is ready
            # (two submits are outstanding at a time on two worker threads: the log's order is not the submits')
            alone = 2 if grid_prefetch is None else 1
            expected = [(1, rows) for rows in pushed[:alone]] + [(len(pushed) - alone, sum(pushed[alone:]))]
            assert sorted(launches) == sorted(expected)


def test_grid_stream_drained_inside_the_library_returns_every_row(hip):
    """bench.py's host_path polls the stream to its end in C++ (mdbh_grid_stream_drain): the rows and
    the first timestamp of every batch must be those the same stream yields batch by batch."""
    timestamps, _, batch = _series(5, length=_n(60_000), irregular=True)
    for batch_size in (8192, 777):
        polled = host.GridStream(hip, batch_size=batch_size)
        drained = host.GridStream(hip, batch_size=batch_size)
        for piece in _segment_batches(batch, {}, 50):
            polled.push(piece)
            drained.push(piece)
        polled.finish_input()
        drained.finish_input()
        batches, state = polled.collect()
        assert state == host.GridStream.READY_NONE
        rows, n_batches, checksum = drained.drain()
        got_ts, _, _ = _concat(batches)
        assert np.array_equal(got_ts, timestamps)
        assert (rows, n_batches) == (len(timestamps), len(batches))
        firsts = sum(int(b.column("timestamp").cast(pa.int64())[0].as_py()) for b in batches)
        assert checksum == firsts % (1 << 64)
        assert polled.metrics()["output_rows"] == drained.metrics()["output_rows"] == len(timestamps)
    points, seconds, bytes_down = host.measure_grid_stream(hip, batch, 8192, segments_per_batch=64)
    assert points == len(timestamps) and bytes_down == 12 * points and seconds > 0


def test_grid_stream_limit_caps_the_batch_size(hip):  # grid_exec.rs:239-246
    _, _, batch = _series(63)
    stream = host.GridStream(hip, tag_names=(), limit=5, batch_size=8192)
    stream.push(batch.to_arrow())
    stream.finish_input()
    state, first = stream.poll_next()
    assert state == host.GridStream.READY_SOME and first.num_rows == 5
    expected = ora.grid_batch(batch)
    assert first.column("timestamp").cast(pa.int64()).to_pylist() == expected[0][:5].tolist()


def test_grid_stream_predicate_prunes_after_reconstruction(hip):  # grid_exec.rs:366-387
    timestamps, _, batch = _series(64, irregular=True)
    lower, upper = int(timestamps[_n(1230)]), int(timestamps[_n(20_000)])
    stream = host.GridStream(hip, tag_names=("tag",), predicate=(lower, upper), batch_size=4096)
    for part in _segment_batches(batch, {"tag": "x"}, 100):
        stream.push(part)
    stream.finish_input()
    batches, _ = stream.collect()
    ts, values, _ = _concat(batches)
    all_ts, all_values, _, _ = ora.grid_batch(batch)
    keep = (all_ts >= lower) & (all_ts <= upper)
    assert np.array_equal(ts, all_ts[keep])
    assert np.array_equal(values.view(np.uint32), all_values[keep].view(np.uint32))


def test_three_point_series_through_the_operators(hip):
    # data_folder.rs:1087-1109, 1165-1234: [37,38,39] / [73,72,71] at three timestamps.
    tags = {"tag": "tag_value"}
    for values, mn, mx, total in (([37.0, 38.0, 39.0], 37.0, 39.0, 114.0),
                                  ([73.0, 72.0, 71.0], 71.0, 73.0, 216.0)):
        segments = host.try_compress_univariate_time_series(hip, [100, 200, 300], values,
                                                            cases.LOSSLESS, tags, 1)
        assert segments.schema.names == list(host.MODEL_SEGMENT_COLUMNS) + ["field_column", "tag"]
        assert segments.column("field_column").to_pylist() == [1] * segments.num_rows
        assert segments.column("tag").to_pylist() == ["tag_value"] * segments.num_rows
        query_batch = segments.drop_columns(["field_column"])
        stream = host.GridStream(hip, tag_names=("tag",))
        stream.push(query_batch)
        stream.finish_input()
        batches, _ = stream.collect()
        ts, reconstructed, table = _concat(batches)
        assert ts.tolist() == [100, 200, 300] and reconstructed.tolist() == values
        assert table.column("tag").to_pylist() == ["tag_value"] * 3
        results = {}
        for name, cls in (("count", host.ModelCountAccumulator), ("min", host.ModelMinAccumulator),
                          ("max", host.ModelMaxAccumulator), ("sum", host.ModelSumAccumulator),
                          ("avg", host.ModelAvgAccumulator)):
            accumulator = cls(hip)
            accumulator.update_batch(query_batch)
            results[name] = accumulator.state()
        assert results["count"] == [3] and results["min"] == [mn] and results["max"] == [mx]
        assert results["sum"] == [total] and results["avg"] == [3, total]


def test_accumulators_over_many_batches_and_state_reset(hip):
    _, _, batch = _series(65)
    parts = _segment_batches(batch, {}, 97)
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    expected = ora.agg_batch(batch, mask)
    accumulators = {name: cls(hip) for name, cls in (
        ("count", host.ModelCountAccumulator), ("min", host.ModelMinAccumulator),
        ("max", host.ModelMaxAccumulator), ("sum", host.ModelSumAccumulator),
        ("avg", host.ModelAvgAccumulator))}
    for part in parts:
        for accumulator in accumulators.values():
            accumulator.update_batch(part)
    assert accumulators["count"].state() == [expected.count]
    assert accumulators["min"].state() == [expected.min]
    assert accumulators["max"].state() == [expected.max]
    total = accumulators["sum"].state()[0]
    assert abs(total - expected.sum) <= 1e-5 * abs(expected.sum)   # integration_test.rs:1155-1171
    count, avg_sum = accumulators["avg"].state()
    assert count == expected.count and abs(avg_sum - expected.sum) <= 1e-5 * abs(expected.sum)
    # state() leaves the accumulator as created (model_simple_aggregates.rs:367-372, 410-415).
    assert accumulators["count"].state() == [0]
    assert accumulators["sum"].state() == [0.0]
    assert accumulators["min"].state() == [np.finfo(np.float32).max]
    assert accumulators["max"].state() == [-np.finfo(np.float32).max]
    assert accumulators["sum"].size() > 0
    for method in ("merge_batch", "evaluate"):                     # unreachable!() in the reference
        with pytest.raises(host.HostError, match="unreachable"):
            getattr(accumulators["sum"], method)()


def test_segment_aggregates_equal_grid_aggregates(hip):  # integration_test.rs:1128-1246
    _, _, batch = _series(66)
    arrow = batch.to_arrow()
    stream = host.GridStream(hip, tag_names=())
    stream.push(arrow)
    stream.finish_input()
    batches, _ = stream.collect()
    _, values, _ = _concat(batches)
    count = host.ModelCountAccumulator(hip)
    minimum, maximum, total = (host.ModelMinAccumulator(hip), host.ModelMaxAccumulator(hip),
                               host.ModelSumAccumulator(hip))
    for accumulator in (count, minimum, maximum, total):
        accumulator.update_batch(arrow)
    assert count.state() == [len(values)]
    assert minimum.state() == [values.min()] and maximum.state() == [values.max()]
    grid_sum = float(values.astype(np.float64).sum())
    assert abs(total.state()[0] - grid_sum) <= 1e-5 * abs(grid_sum)


def test_try_compress_univariate_error_behaviour(hip):
    empty = host.try_compress_univariate_time_series(hip, [], [], cases.LOSSLESS, {"tag": "t"}, 0)
    assert empty.num_rows == 0                                                        # :422-434
    assert empty.schema.names == list(host.MODEL_SEGMENT_COLUMNS) + ["field_column", "tag"]
    with pytest.raises(host.HostError, match="different lengths"):                   # :202-206
        host.try_compress_univariate_time_series(hip, [1, 2], [1.0], cases.LOSSLESS, {}, 0)
    with pytest.raises(host.HostError, match="positive finite"):   # ErrorBound::try_new_absolute
        host.try_compress_univariate_time_series(hip, [1], [1.0], mdb._abi.ErrorBoundC(1, -1.0), {}, 0)
    with pytest.raises(host.HostError, match="at most 100.0%"):    # ErrorBound::try_new_relative
        host.try_compress_univariate_time_series(hip, [1], [1.0], mdb._abi.ErrorBoundC(2, 101.0), {}, 0)


def test_try_compress_univariate_known_segment(hip):  # compression.rs:932-978
    segments = host.try_compress_univariate_time_series(
        hip, [100, 200, 300, 400, 500], [73.0, 37.0, 37.0, 37.0, 73.0], cases.LOSSLESS, {"tag": "tag"}, 0)
    assert segments.num_rows == 1
    row = {name: segments.column(name)[0].as_py() for name in segments.schema.names}
    assert row["model_type_id"] == 2
    assert segments.column("start_time").cast(pa.int64())[0].as_py() == 100
    assert segments.column("end_time").cast(pa.int64())[0].as_py() == 500
    assert row["timestamps"] == bytes([5]) and (row["min_value"], row["max_value"]) == (37.0, 73.0)
    assert len(row["values"]) == 8 and row["residuals"] == b"" and np.isnan(row["error"])


def test_try_compress_multivariate_sorts_splits_and_compresses(hip):  # compression.rs:42-179
    rng = np.random.default_rng(71)
    n = 4000
    series = {("A", "x"): 81, ("B", "a-long-tag-value-beyond-12-bytes"): 82, ("A", "y"): 83}
    columns = {"timestamp": [], "field_1": [], "field_2": [], "tag_1": [], "tag_2": []}
    per_series = {}
    for (tag_1, tag_2), seed in series.items():
        ts, v1 = cases.synthetic_series(n, False, (1.0, 1.05), seed)
        _, v2 = datagen.sine_series(seed, n)
        per_series[(tag_1, tag_2)] = (ts, v1, v2)
        columns["timestamp"].append(ts)
        columns["field_1"].append(v1)
        columns["field_2"].append(v2)
        columns["tag_1"] += [tag_1] * n
        columns["tag_2"] += [tag_2] * n
    order = rng.permutation(3 * n)                                       # unsorted ingest order
    batch = pa.RecordBatch.from_arrays([
        pa.array(np.concatenate(columns["timestamp"])[order], type=pa.int64()).cast(pa.timestamp("us")),
        pa.array(np.concatenate(columns["field_1"])[order], type=pa.float32()),
        pa.array(np.array(columns["tag_1"])[order], type=pa.string_view()),
        pa.array(np.concatenate(columns["field_2"])[order], type=pa.float32()),
        pa.array(np.array(columns["tag_2"])[order], type=pa.string_view()),
    ], names=["timestamp", "field_1", "tag_1", "field_2", "tag_2"])
    bounds = {1: mdb.error_bound("relative", 5.0), 3: mdb.error_bound("lossless")}
    out = host.try_compress_multivariate_time_series(hip, batch, 0, [1, 3], [2, 4], bounds)
    assert len(out) == len(series) * 2
    expected_order = sorted(series)                                       # lexsort by tags
    for s, key in enumerate(expected_order):
        ts, v1, v2 = per_series[key]
        for f, (field_index, values) in enumerate(((1, v1), (3, v2))):
            got = out[s * 2 + f]
            assert got.column("field_column").to_pylist() == [field_index] * got.num_rows
            assert got.column("tag_1").to_pylist() == [key[0]] * got.num_rows
            assert got.column("tag_2").to_pylist() == [key[1]] * got.num_rows
            expected = ora.try_compress_univariate_time_series(ts, values, bounds[field_index])
            got_segments = mdb.SegmentBatch.from_arrow(got)
            assert got_segments.rows() == expected.rows()


def test_uncompressed_data_manager_compresses_finished_buffers_in_one_launch(hip):
    # N4: many series' finished buffers x fields -> one mdb_compress_chunks call per error bound;
    # the result must equal the reference's per-buffer, per-field compression of the time-sorted
    # buffer (uncompressed_data_manager.rs:530-596, uncompressed_data_buffer.rs:175-209).
    rng = np.random.default_rng(111)
    n_series, n_points, capacity = 5, 3000, 1024
    data = {}
    rows = []
    for s in range(n_series):
        ts, v1 = cases.synthetic_series(n_points, s % 2 == 1, (1.0, 1.05), 600 + s)
        _, v2 = datagen.sine_series(s, n_points)
        data[f"series-{s}"] = (ts, v1, v2)
    # ingest batches of 500 points per series
    bounds = {1: mdb.error_bound("relative", 5.0), 2: mdb.error_bound("relative", 5.0)}
    manager = None
    expected = []
    for start in range(0, n_points, 500):
        cols = {"timestamp": [], "field_1": [], "field_2": [], "tag": []}
        for tag, (ts, v1, v2) in data.items():
            cols["timestamp"].append(ts[start:start + 500])
            cols["field_1"].append(v1[start:start + 500])
            cols["field_2"].append(v2[start:start + 500])
            cols["tag"] += [tag] * 500
        # Series interleaved row by row (each series stays in time order, as a sensor would send it).
        order = np.arange(500 * n_series).reshape(n_series, 500).T.ravel()
        batch = pa.RecordBatch.from_arrays([
            pa.array(np.concatenate(cols["timestamp"])[order], type=pa.int64()).cast(pa.timestamp("us")),
            pa.array(np.concatenate(cols["field_1"])[order], type=pa.float32()),
            pa.array(np.concatenate(cols["field_2"])[order], type=pa.float32()),
            pa.array(np.array(cols["tag"])[order], type=pa.string_view())],
            names=["timestamp", "field_1", "field_2", "tag"])
        if manager is None:
            manager = host.UncompressedDataManager(hip, batch.schema, 0, [1, 2], [3], bounds,
                                                   buffer_capacity=capacity)
        manager.insert_data_points(batch)
    manager.flush()
    active, finished = manager.counts()
    assert active == 0 and finished == n_series * ((n_points + capacity - 1) // capacity)
    compressed = manager.compress_finished_buffers()
    assert len(compressed) == finished * 2
    assert manager.counts() == (0, 0)
    # Every (series, field): the concatenation of its buffers' segments reconstructs the series.
    per_key = {}
    for batch in compressed:
        if batch.num_rows == 0:
            continue
        key = (batch.column("tag")[0].as_py(), batch.column("field_column")[0].as_py())
        per_key.setdefault(key, []).append(batch)
    assert len(per_key) == n_series * 2
    for (tag, field), batches in per_key.items():
        ts, v1, v2 = data[tag]
        values = v1 if field == 1 else v2
        segments = mdb.SegmentBatch.concat(
            [mdb.SegmentBatch.from_arrow(b) for b in sorted(
                batches, key=lambda b: b.column("start_time").cast(pa.int64())[0].as_py())])
        got_ts, got_values, _, _ = ora.grid_batch(segments)
        assert np.array_equal(got_ts, ts)
        within = np.abs((values - got_values) / values) * np.float32(100.0) <= np.float32(5.0)
        assert within.all() or np.isclose(values[~within], got_values[~within]).all()
        # and buffer by buffer it equals the oracle's compression of that time-sorted buffer
        for b in batches:
            first = b.column("start_time").cast(pa.int64())[0].as_py()
            last = b.column("end_time").cast(pa.int64())[b.num_rows - 1].as_py()
            lo, hi = np.searchsorted(ts, first), np.searchsorted(ts, last) + 1
            oracle = ora.try_compress_univariate_time_series(ts[lo:hi], values[lo:hi], bounds[field])
            assert mdb.SegmentBatch.from_arrow(b).rows() == oracle.rows()


@pytest.mark.skipif(UNDER_STUB, reason="device-resident path: needs the GPU")
def test_segment_files_to_device_and_grid(hip, tmp_path):
    # N2: whole Parquet segment files -> one device batch -> grid, equal to the oracle.
    from modelardb_rs_amd import segment_files
    _, _, batch = _series(67)
    arrow = host.segments_with_tags(batch.to_arrow(), {"tag": "T"})
    paths = [segment_files.write_segment_file(str(tmp_path / f"part-{k}.parquet"), part)
             for k, part in enumerate((arrow.slice(0, 100), arrow.slice(100)))]
    device_segments, tags = segment_files.load_segments(hip, paths)
    assert tags.column("tag").to_pylist() == ["T"] * len(batch)
    total = hip.grid_count_dev(device_segments)
    out_ts, out_val = hip.dev_alloc(8 * total), hip.dev_alloc(4 * total)
    hip.grid_batch_dev(device_segments, out_ts, out_val, total)
    expected = ora.grid_batch(batch)
    assert np.array_equal(hip.download_array(out_ts, total, np.int64), expected[0])
    assert np.array_equal(hip.download_array(out_val, total, np.float32).view(np.uint32),
                          expected[1].view(np.uint32))
    # ... and row group after row group (decoded by threads while the one before is uploaded and reconstructed):
    # the same points, each group's behind the last one's
    done = 0
    for group, group_tags in segment_files.load_segments_pipelined(hip, paths, workers=2):
        n = hip.grid_count_dev(group)
        group_ts, group_val = hip.dev_alloc(8 * n), hip.dev_alloc(4 * n)
        hip.grid_batch_dev(group, group_ts, group_val, n)
        assert group_tags.column("tag").to_pylist() == ["T"] * len(group)
        assert len(group.download()) == len(group)  # (the batch is the caller's context's: the uploader's is closed by then or soon)
        assert np.array_equal(hip.download_array(group_ts, n, np.int64), expected[0][done:done + n])
        assert np.array_equal(hip.download_array(group_val, n, np.float32).view(np.uint32),
                              expected[1][done:done + n].view(np.uint32))
        done += n
        hip.dev_free(group_ts)
        hip.dev_free(group_val)
        group.free()
    assert done == total
    hip.dev_free(out_ts)
    hip.dev_free(out_val)
    device_segments.free()


@pytest.mark.parametrize("predicate", [(None, None), "middle"])
def test_sorted_join_of_three_field_columns(hip, predicate, monkeypatch):
    # sorted_join_exec.rs:277-311 over one GridExec per field column: timestamps and
    # tags come from the first field's GridExec, the other two only reconstruct values.
    # (ragged pushes make the inputs' batches differ in length: with the surplus carried over the join returns
    # every point row-aligned, which is what can be checked against the oracle - by default the surplus is
    # dropped as the reference drops it, tests/test_host_ops_cpu.py)
    monkeypatch.setenv("MDB_HOST_SORTED_JOIN_CARRY_OVER", "1")
    eb = cases.error_bounds()["rel5"]
    timestamps, _ = cases.synthetic_series(_n(40_000), True, (1.0, 1.05), 90)
    if predicate == "middle":
        predicate = (int(timestamps[_n(5_000)]), int(timestamps[_n(31_230)]))
    fields = []
    for seed in (91, 92, 93):
        _, values = cases.synthetic_series(_n(40_000), True, (1.0, 1.05), seed)
        fields.append(ora.try_compress_univariate_time_series(timestamps, values, eb))
    order = ["timestamp", "field", "field", ("tag", "tag"), "field"]
    join = host.SortedJoinStream(hip, 3, order, tag_names=("tag",), predicate=predicate, batch_size=_n(4096))
    assert "values_only=011" in join.describe()
    for index, batch in enumerate(fields):
        for part in _segment_batches(batch, {"tag": "wind-turbine-1234567"}, 37 + index):  # ragged on purpose
            join.push(index, part)
    join.finish_input()
    batches, state = join.collect()
    assert state == host.SortedJoinStream.READY_NONE
    table = pa.Table.from_batches(batches)
    assert table.schema.names == ["timestamp", "field_0", "field_1", "tag", "field_2"]
    expected = [ora.grid_batch(batch) for batch in fields]
    keep = np.ones(len(expected[0][0]), dtype=bool)
    if predicate != (None, None):
        keep = (expected[0][0] >= predicate[0]) & (expected[0][0] <= predicate[1])
    # The inputs emit batches of different sizes (ragged pushes, short batches under the predicate);
    # the join keeps them row-aligned and returns every point.
    n = table.num_rows
    assert n == int(keep.sum())
    assert np.array_equal(table.column("timestamp").cast(pa.int64()).to_numpy(), expected[0][0][keep][:n])
    for column, (_, values, _, _) in zip(("field_0", "field_1", "field_2"), expected):
        assert np.array_equal(table.column(column).to_numpy().view(np.uint32), values[keep][:n].view(np.uint32))
    assert set(table.column("tag").to_pylist()) == {"wind-turbine-1234567"}


# ---- SURVEY 8(f) N1 through the operators: a range on the timestamp pushed into the library by GridStream
# (rust/patches/0001: time_range_of_predicate) and by the accumulators the extended ModelSimpleAggregates rule puts in
# (rust/patches/0002), against grid + filter + aggregate - the plan the reference runs for such a query.

def _stub_calls():
    log = os.environ.get("MDB_STUB_CALL_LOG")
    if not log or not os.path.exists(log):
        return None
    signed = lambda text: int(text) - (1 << 64) if int(text) >= (1 << 63) else int(text)
    return [(what, signed(a), signed(b)) for what, a, b, _ in (line.split() for line in open(log))]


def _clear_stub_calls():
    log = os.environ.get("MDB_STUB_CALL_LOG")
    if log and os.path.exists(log):
        os.remove(log)


@pytest.mark.parametrize("shape", ["inclusive", "strict", "literal-left", "one-sided", "equal", "nothing"])
def test_grid_stream_pushes_the_range_of_its_predicate_into_the_library(hip, shape):
    timestamps, _, batch = _series(68, irregular=True)
    lower, upper = int(timestamps[_n(1230)]), int(timestamps[_n(20_000)])
    predicate, keep_lower, keep_upper = {
        "inclusive": (f"(and (>= timestamp ts:{lower}) (<= timestamp ts:{upper}))", lower, upper),
        "strict": (f"(and (> timestamp ts:{lower}) (< timestamp ts:{upper}))", lower + 1, upper - 1),
        "literal-left": (f"(and (<= ts:{lower} timestamp) (> ts:{upper} timestamp))", lower, upper - 1),
        "one-sided": (f"(> timestamp ts:{upper})", upper + 1, (1 << 63) - 1),
        "equal": (f"(= timestamp ts:{lower})", lower, lower),
        "nothing": (f"(and (>= timestamp ts:{upper}) (< timestamp ts:{lower}))", (1 << 63) - 1, -(1 << 63)),
    }[shape]
    assert host.time_range_of_predicate(predicate) == (keep_lower, keep_upper, True)
    _clear_stub_calls()
    stream = host.GridStream(hip, tag_names=("tag",), predicate=predicate, batch_size=4096)
    for part in _segment_batches(batch, {"tag": "x"}, 100):
        stream.push(part)
    stream.finish_input()
    batches, state = stream.collect()
    assert state == host.GridStream.READY_NONE
    all_ts, all_values, _, _ = ora.grid_batch(batch)
    keep = (all_ts >= keep_lower) & (all_ts <= keep_upper)
    if shape == "nothing":
        assert not keep.any() and sum(b.num_rows for b in batches) == 0
    else:
        ts, values, _ = _concat(batches)
        assert np.array_equal(ts, all_ts[keep])
        assert np.array_equal(values.view(np.uint32), all_values[keep].view(np.uint32))
    # only the points inside the range were reconstructed at all (grid_exec.rs:366-387 reconstructs every point)
    assert stream.metrics()["rows_created"] == int(keep.sum())
    calls = _stub_calls()
    if calls is not None:   # (under tests/stub: what the kernels' side saw)
        assert [(lo, hi) for what, lo, hi in calls if what == "grid_range"] \
            == [(keep_lower, keep_upper)] * len([1 for what, _, _ in calls if what == "grid"])


def test_grid_stream_filters_behind_the_library_what_the_range_does_not_decide(hip):
    timestamps, _, batch = _series(69, irregular=True)
    lower, a, b = int(timestamps[_n(1000)]), int(timestamps[_n(9_000)]), int(timestamps[_n(15_000)])
    # a range and a hole in it: the range is pushed down, the OR is evaluated on what comes back
    predicate = f"(and (>= timestamp ts:{lower}) (or (< timestamp ts:{a}) (> timestamp ts:{b})))"
    assert host.time_range_of_predicate(predicate) == (lower, (1 << 63) - 1, False)
    for text in (predicate, f"(or (< timestamp ts:{a}) (> timestamp ts:{b}))"):   # (the second: nothing to push)
        stream = host.GridStream(hip, tag_names=("tag",), predicate=text, batch_size=1000)
        for part in _segment_batches(batch, {"tag": "a-tag-value-longer-than-12-bytes"}, 64):
            stream.push(part)
        stream.finish_input()
        batches, _ = stream.collect()
        ts, values, table = _concat(batches)
        all_ts, all_values, _, _ = ora.grid_batch(batch)
        keep = (all_ts < a) | (all_ts > b)
        if text is predicate:
            keep &= all_ts >= lower
        assert np.array_equal(ts, all_ts[keep])
        assert np.array_equal(values.view(np.uint32), all_values[keep].view(np.uint32))
        assert set(table.column("tag").to_pylist()) == {"a-tag-value-longer-than-12-bytes"}


def test_accumulators_under_a_time_range(hip):
    timestamps, _, batch = _series(70, irregular=True)
    lower, upper = int(timestamps[_n(4_321)]), int(timestamps[_n(22_222)])
    parts = _segment_batches(batch, {}, 97)
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    expected = ora.agg_batch_range(batch, lower, upper, mask)
    _clear_stub_calls()
    accumulators = {name: cls(hip, (lower, upper)) for name, cls in (
        ("count", host.ModelCountAccumulator), ("min", host.ModelMinAccumulator),
        ("max", host.ModelMaxAccumulator), ("sum", host.ModelSumAccumulator),
        ("avg", host.ModelAvgAccumulator))}
    for part in parts:
        for accumulator in accumulators.values():
            accumulator.update_batch(part)
    assert accumulators["count"].state() == [expected.count]
    assert accumulators["min"].state() == [expected.min]
    assert accumulators["max"].state() == [expected.max]
    total = accumulators["sum"].state()[0]
    assert abs(total - expected.sum) <= 1e-5 * abs(expected.sum)   # integration_test.rs:1155-1171
    count, avg_sum = accumulators["avg"].state()
    assert count == expected.count and abs(avg_sum - expected.sum) <= 1e-5 * abs(expected.sum)
    calls = _stub_calls()
    if calls is not None:   # ONE ranged call per accumulator for all its batches (PendingSegments::fold_into)
        assert [n for what, n, _ in calls if what == "agg_range_list"] == [len(parts)] * 5
        assert {(lo, hi) for what, lo, hi in calls if what == "agg_range"} == {(lower, upper)}
        assert not [1 for what, _, _ in calls if what == "agg_list"]
    # a range without a data point: COUNT 0, the others NULL (DataFusion's MIN / MAX / SUM over no rows)
    empty = (int(timestamps[-1]) + 1, int(timestamps[-1]) + 1000)
    for name, cls in (("count", host.ModelCountAccumulator), ("min", host.ModelMinAccumulator),
                      ("max", host.ModelMaxAccumulator), ("sum", host.ModelSumAccumulator)):
        accumulator = cls(hip, empty)
        accumulator.update_batch(parts[-1])
        assert accumulator.state() == ([0] if name == "count" else [None])
    accumulator = host.ModelAvgAccumulator(hip, empty)
    accumulator.update_batch(parts[-1])
    assert accumulator.state() == [0, 0.0]   # (DataFusion's AVG is NULL for a count of zero whatever the sum)


@pytest.mark.parametrize("irregular", [False, True], ids=["regular", "irregular"])
def test_aggregate_query_under_a_time_range_through_the_rule(hip, irregular):
    """SELECT COUNT, MIN, MAX, SUM (and AVG) of a field WHERE lower <= timestamp < upper: the plan the reference runs
    (DataSourceExec -> GridExec -> SortedJoinExec -> FilterExec -> AggregateExec) and the plan the extended rule makes
    of it (DataSourceExec -> AggregateExec over the segments) give the same answer, which is the oracle's."""
    query = host.AggregateQuery(hip, n_fields=1, tag_names=("tag",))
    batches, last = [], 0
    for seed, tag in ((71, "A"), (72, "B")):
        timestamps, _, batch = _series(seed, irregular=irregular)
        batches.append(batch)
        last = max(last, int(timestamps[-1]))
        for part in _segment_batches(batch, {"tag": tag}, 113):
            query.push_segments(0, part)
    lower, upper = int(timestamps[_n(3_000)]), int(timestamps[_n(17_000)])
    filters = [f"(>= timestamp ts:{lower})", f"(< timestamp ts:{upper})"]
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    expected = ora.agg_batch_range(batches[1], lower, upper - 1, mask, ora.agg_batch_range(batches[0], lower, upper - 1, mask))
    aggregates = [("count", 0), ("min", 0), ("max", 0), ("sum", 0)]
    results = {}
    for optimize in (False, True):
        query.plan(aggregates, filters, optimize=optimize)
        assert (query.levels()[3] == ["DataSourceExec"]) == optimize
        results[optimize] = query.execute(batch_size=_n(8192))
    for count, minimum, maximum, total in results.values():
        assert count == expected.count and minimum == expected.min and maximum == expected.max
        assert abs(total - expected.sum) <= 1e-5 * abs(expected.sum)
    query.plan([("avg", 0)], filters, optimize=True)
    assert query.aggregates() == [f"model_avg[{lower},{upper - 1}]"]
    (average,) = query.execute()
    assert abs(average - expected.sum / expected.count) <= 1e-5 * abs(expected.sum / expected.count)
    # nothing in the range: COUNT 0 and NULLs from both plans
    beyond = [f"(> timestamp ts:{last + 10})"]
    for optimize in (False, True):
        assert query.plan(aggregates, beyond, optimize=optimize).execute() == [0.0, None, None, None]
    # without a predicate the rule's original case: the segments' own metadata (model_simple_aggregates.rs:336-618)
    whole = ora.agg_batch(batches[1], mask, ora.agg_batch(batches[0], mask))
    count, minimum, maximum, total = query.plan(aggregates, (), optimize=True).execute()
    assert count == whole.count and minimum == whole.min and maximum == whole.max
    assert abs(total - whole.sum) <= 1e-5 * abs(whole.sum)
