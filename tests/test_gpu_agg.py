"""Parity of the HIP segment aggregates with the CPU oracle, through the C ABI.

Bar (from the reference's own tests, crates/modelardb_server/tests/integration_test.rs:1128-1246):
COUNT / MIN / MAX exact; SUM / AVG within 0.001 % relative (the GPU reduces in a fixed tree, the
reference accumulates sequentially in f64)."""

import os

import numpy as np
import pytest

import cases
import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import MDB_AGG_AVG, MDB_AGG_COUNT, MDB_AGG_MAX, MDB_AGG_MIN, MDB_AGG_SUM

pytestmark = pytest.mark.gpu

ALL = MDB_AGG_COUNT | MDB_AGG_MIN | MDB_AGG_MAX | MDB_AGG_SUM
SUM_TOLERANCE = 1e-5  # 0.001 %


@pytest.fixture(autouse=True, params=[None, "off", "8", "no-walk", "host-cursors"],
                ids=["mv-default", "mv-off", "mv-from-8-values", "timestamps-every-lane-for-itself",
                     "mv-cursors-by-host-threads"])
def macaque_decoder(request, monkeypatch):
    """SUM leaves long MacaqueV streams to the parallel decoder (macaque_deferred_sum in mdb_grid.hip):
    every test runs with its default threshold, with it switched off and with every stream of at
    least 8 values going that way. len() and swing::sum of segments with irregular timestamps come from the
    wave-synchronous walk of their streams (k_grid_ts_count<SUMS>), or - the last mode - from every lane
    decoding its own stream."""
    monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)
    monkeypatch.delenv("MDB_GRID_MV_HOST_MIN_VALUES", raising=False)
    if request.param is None:
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
    elif request.param == "host-cursors":
        # SUM over a host batch: the call's host threads walk the MacaqueV streams of segments with regular timestamps
        # (by default the long ones, here all) and the values are decoded piece by piece from their cursors.
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
        monkeypatch.setenv("MDB_GRID_MV_HOST_MIN_VALUES", "1")
    elif request.param == "no-walk":
        monkeypatch.delenv("MDB_GRID_MV_MIN_VALUES", raising=False)
        monkeypatch.setenv("MDB_AGG_TS_WALK", "0")
    else:
        monkeypatch.setenv("MDB_GRID_MV_MIN_VALUES", request.param)
    return request.param


def _assert_state(got, expected):
    assert got.count == expected.count
    assert np.float32(got.min) == np.float32(expected.min)
    assert np.float32(got.max) == np.float32(expected.max)
    if np.isnan(expected.sum) or np.isinf(expected.sum):
        assert np.isnan(got.sum) == np.isnan(expected.sum)
        assert np.isinf(got.sum) == np.isinf(expected.sum)
    else:
        assert abs(got.sum - expected.sum) <= SUM_TOLERANCE * max(abs(expected.sum), 1e-30)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("eb_name", ["lossless", "abs5", "rel5", "rel1"])
def test_aggregates_match_oracle(hip, eb_name, irregular):
    eb = cases.error_bounds()[eb_name]
    _, _, batch = cases.mixed_batch(eb, irregular, seed=21)
    _assert_state(hip.agg_batch(batch, ALL), ora.agg_batch(batch, ALL))
    for mask in (MDB_AGG_COUNT, MDB_AGG_MIN, MDB_AGG_MAX, MDB_AGG_SUM, MDB_AGG_AVG):
        got, expected = hip.agg_batch(batch, mask), ora.agg_batch(batch, mask)
        _assert_state(got, expected)


@pytest.mark.parametrize("eb_name", ["lossless", "abs5", "rel5"])
@pytest.mark.parametrize("which", ["count", "min", "max", "sum", "avg"])
def test_aggregate_from_segments_equals_aggregate_from_data_points(hip, which, eb_name):
    # crates/modelardb_server/tests/integration_test.rs:1128-1171, 1173-1246: the same query answered
    # from the segments (the optimizer rule) and from the data points reconstructed from them
    # (GridExec + AggregateExec) - COUNT, MIN and MAX equal, SUM and AVG within 0.001 %. Here both
    # sides are the library's: mdb_agg_batch against numpy over mdb_grid_batch. Regular timestamps,
    # as in the reference's test: with irregular ones swing::sum draws its line to the SEGMENT's end
    # time while grid() draws it to the model's (swing.rs:264-300 vs models/mod.rs:219-234), so a Swing
    # segment with a residual tail sums to something else than its points - up to 0.9 % in
    # mixed_batch(lossless, irregular) - and the library follows the reference there (the oracle
    # tests above), not this identity.
    eb = cases.error_bounds()[eb_name]
    _, _, batch = cases.mixed_batch(eb, False, seed=27)
    _, points, _, _ = hip.grid_batch(batch)
    mask = {"count": MDB_AGG_COUNT, "min": MDB_AGG_MIN, "max": MDB_AGG_MAX, "sum": MDB_AGG_SUM, "avg": MDB_AGG_AVG}[which]
    state = hip.agg_batch(batch, mask)
    if which == "count":
        assert state.count == len(points)
    elif which == "min":
        assert np.float32(state.min) == points.min()
    elif which == "max":
        assert np.float32(state.max) == points.max()
    else:
        total = float(points.astype(np.float64).sum())
        expected = total if which == "sum" else total / len(points)
        got = state.sum if which == "sum" else state.sum / state.count
        assert abs(got - expected) <= 0.001 / 100.0 * abs(expected)


def test_aggregates_continue_a_running_state(hip):
    eb = cases.error_bounds()["rel5"]
    _, _, batch = cases.mixed_batch(eb, False, seed=22)
    half = len(batch) // 2
    first, second = batch.slice(0, half), batch.slice(half, len(batch))
    state = hip.agg_batch(first, ALL)
    state = hip.agg_batch(second, ALL, state)
    _assert_state(state, ora.agg_batch(batch, ALL))


def test_a_list_of_batches_is_folded_as_one(hip):
    # mdb_agg_batch_list: what an accumulator that has gathered the batches of a run of update_batch calls passes.
    # COUNT / MIN / MAX as the batches one by one, SUM to the rounding of the f64 additions.
    batches = []
    for eb_name, irregular, seed in (("lossless", False, 3), ("rel1", True, 4), ("abs5", False, 5), ("rel5", True, 6)):
        batches.append(cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=seed)[2])
    one_by_one = None
    for batch in batches:
        one_by_one = hip.agg_batch(batch, ALL, one_by_one)
    together = hip.agg_batch_list(batches, ALL)
    assert (together.count, together.min, together.max) == (one_by_one.count, one_by_one.min, one_by_one.max)
    assert abs(together.sum - one_by_one.sum) <= 1e-12 * abs(one_by_one.sum)
    expected = ora.agg_batch(batches[0], ALL)
    for batch in batches[1:]:
        expected = ora.agg_batch(batch, ALL, expected)
    _assert_state(together, expected)
    assert hip.agg_batch_list([], ALL).count == 0
    seeded = hip.agg_batch_list(batches[:1], ALL, hip.agg_batch_list(batches[1:], ALL))
    assert seeded.count == together.count and abs(seeded.sum - together.sum) <= 1e-12 * abs(together.sum)


def test_sum_of_long_lossless_streams(hip, macaque_decoder, monkeypatch):
    # BASELINE configs[0] as an aggregate query: 16 MacaqueV streams of 65 536 values. Each stream is
    # added up in f32 in stream order whoever decodes it (macaque_v.rs:220-265), so the two ways must
    # agree to the rounding of a handful of f64 additions.
    import datagen
    n = 1_000_000
    timestamps, values = datagen.sine_series(3, n)
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
    offsets[-1] = n
    segments = hip.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    assert set(segments.model_type_id.tolist()) == {2}
    # (with the cursors of the call's host threads the values come from the piece decoder and one lane adds a stream
    # up; without them - MDB_GRID_MV_INDEX=0 - from the parallel decoder unless that is switched off)
    for index in ("on", "off"):
        if index == "off":
            monkeypatch.setenv("MDB_GRID_MV_INDEX", "0")
        hip.profile_enable(True)
        hip.profile_reset()
        state = hip.agg_batch(segments, ALL)
        kernels = hip.profile()
        hip.profile_enable(False)
        if index == "on":
            assert "k_agg_mv_chains" in kernels and "k_mv_sums" not in kernels
        else:
            assert ("k_mv_sums" in kernels) == (macaque_decoder != "off")
        _assert_state(state, ora.agg_batch(segments, ALL))
    monkeypatch.delenv("MDB_GRID_MV_INDEX")
    per_stream = 0.0
    for k in range(len(offsets) - 1):
        per_stream += float(np.add.accumulate(values[int(offsets[k]):int(offsets[k + 1])], dtype=np.float32)[-1])
    assert abs(state.sum - per_stream) <= 1e-12 * abs(per_stream)
    # The same under a time range (what the reference computes with GridExec + filter + aggregate):
    # the values with index 300 000 .. 700 000, which lie in 8 of the 16 streams.
    # With the host threads' cursors: the pieces that reach into the range (k_agg_mv_range); without: the parallel decoder.
    lo, hi = 300_000, 700_000
    inside = values[lo:hi + 1]
    for index in ("on", "off"):
        if index == "off":
            monkeypatch.setenv("MDB_GRID_MV_INDEX", "0")
        hip.profile_enable(True)
        hip.profile_reset()
        ranged = hip.agg_batch_range(segments, int(timestamps[lo]), int(timestamps[hi]), ALL)
        kernels = hip.profile()
        hip.profile_enable(False)
        if index == "on":
            assert "k_agg_mv_range" in kernels and "k_mv_range_partials" not in kernels
        else:
            assert "k_agg_mv_range" not in kernels and ("k_mv_range_finish" in kernels) == (macaque_decoder != "off")
        assert (ranged.count, ranged.min, ranged.max) == (len(inside), inside.min(), inside.max())
        assert abs(ranged.sum - float(inside.astype(np.float64).sum())) <= 1e-9 * abs(float(inside.astype(np.float64).sum()))
    monkeypatch.delenv("MDB_GRID_MV_INDEX")
    monkeypatch.setenv("MDB_GRID_MV_MIN_VALUES", "off")
    serial = hip.agg_batch(segments, ALL)
    assert abs(state.sum - serial.sum) <= 1e-12 * abs(serial.sum)
    assert (state.count, state.min, state.max) == (serial.count, serial.min, serial.max)


def test_aggregates_three_point_series(hip):
    # crates/modelardb_embedded/src/operations/data_folder.rs:1165-1234
    for values, mn, mx, total in (([37.0, 38.0, 39.0], 37.0, 39.0, 114.0),
                                  ([73.0, 72.0, 71.0], 71.0, 73.0, 216.0)):
        batch = ora.try_compress_univariate_time_series([100, 200, 300], values, cases.LOSSLESS)
        state = hip.agg_batch(batch, ALL)
        assert (state.count, state.min, state.max, state.sum) == (3, mn, mx, total)


def test_aggregates_edge_cases(hip):
    for name, ts, values in cases.edge_case_series():
        batch = ora.try_compress_univariate_time_series(ts, values, cases.LOSSLESS)
        _assert_state(hip.agg_batch(batch, ALL), ora.agg_batch(batch, ALL))


def test_fuzzed_segments_never_hang_and_agree_with_the_oracle(hip):
    """Valid segments with random corruptions (truncated or random payloads, shifted times, wrong
    model type), as in the grid test of the same name: an error or a result, never a crash or a
    hang - a corrupted length can claim 2^31 values for a 12-byte stream - and whenever the oracle
    accepts a batch (for aggregates and for grid) the GPU accepts it too and agrees."""
    rng = np.random.default_rng(231)
    pool = []
    for eb_name in ("lossless", "rel5"):
        pool += cases.edge_case_batch(cases.error_bounds()[eb_name]).rows()
        for irregular in (False, True):
            pool += cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=232, length=3000)[2].rows()
    agree = errors = 0
    for trial in range(300):
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
            expected = ora.agg_batch(batch, ALL)
            # len() and sum() look at less of a segment than grid() does (a regular segment whose
            # length does not fit its time span counts and sums fine in the reference and panics in
            # grid()); the library validates a segment the same way for both, so only batches that
            # grid() accepts as well have to be accepted here.
            if expected.count > 200_000:
                continue
            ora.grid_batch(batch)
        except ora.OracleError:
            expected = None
        try:
            got = hip.agg_batch(batch, ALL)
        except mdb.HipError:
            got = None
        if expected is not None:
            assert got is not None, rows
            _assert_state(got, expected)
            agree += 1
        else:
            errors += got is None
    assert agree > 50 and errors > 20, (agree, errors)


def test_a_stream_shorter_than_its_segment_claims_is_an_error_not_a_hang(hip):
    # 2^31 - 1 values according to the timestamps column, 8 bytes of MacaqueV values.
    good = ora.try_compress_univariate_time_series([100, 200, 300, 400, 500], [73.0, 37.0, 37.0, 37.0, 73.0], cases.LOSSLESS)
    row = list(good.rows()[0])
    assert row[0] == 2
    count = (1 << 31) - 1
    row[2] = row[1] + (count - 1) * 100
    row[3] = count.to_bytes(4, "big")
    batch = mdb.SegmentBatch.from_rows([tuple(row)])
    with pytest.raises(ora.OracleError):
        ora.agg_batch(batch, ALL)
    with pytest.raises(mdb.HipError, match="MacaqueV"):
        hip.agg_batch(batch, ALL)
    assert hip.agg_batch(batch, MDB_AGG_COUNT).count == count  # len() never looks at the values


def test_a_regular_segment_that_ends_before_it_starts(hip):
    # (start..=end).step_by(..) is empty (timestamps.rs:218-222), so grid() yields nothing, while len()
    # reports the stored length and pmc_mean::sum multiplies by it: the library follows both.
    batch = mdb.SegmentBatch.from_rows([(0, 280, 70, b"\x08", 5.0, 5.0, b"", b"")])
    expected = ora.agg_batch(batch, ALL)
    assert (expected.count, expected.sum) == (8, 40.0)
    _assert_state(hip.agg_batch(batch, ALL), expected)
    assert len(ora.grid_batch(batch)[0]) == 0
    assert len(hip.grid_batch(batch)[0]) == 0


def test_aggregates_empty_batch(hip):
    state = hip.agg_batch(mdb.SegmentBatch.from_rows([]), ALL)
    fresh = mdb._abi.AggStateC.fresh()
    assert (state.count, state.sum, state.min, state.max) == (0, 0.0, fresh.min, fresh.max)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("eb_name", ["lossless", "rel5", "abs5"])
def test_range_aggregates_match_grid_filter_aggregate(hip, eb_name, irregular):
    # The reference evaluates WHERE timestamp BETWEEN lo AND hi as GridExec + filter + aggregate
    # (model_simple_aggregates.rs:284-302); ora.agg_batch_range is that plan.
    eb = cases.error_bounds()[eb_name]
    timestamps, _, batch = cases.mixed_batch(eb, irregular, seed=23)
    n = len(timestamps)
    windows = [
        (int(timestamps[n // 4]), int(timestamps[3 * n // 4])),
        (int(timestamps[0]), int(timestamps[-1])),
        (int(timestamps[10]) + 1, int(timestamps[11]) - 1) if timestamps[11] - timestamps[10] > 1
        else (int(timestamps[10]), int(timestamps[10])),
        (int(timestamps[100]), int(timestamps[100])),
        (int(timestamps[-1]) + 1, int(timestamps[-1]) + 1000),
        (-(1 << 62), 1 << 62),
        (int(timestamps[n // 2]) - 37, int(timestamps[n // 2]) + 4242),
    ]
    for t_lo, t_hi in windows:
        got = hip.agg_batch_range(batch, t_lo, t_hi, ALL)
        expected = ora.agg_batch_range(batch, t_lo, t_hi, ALL)
        _assert_state(got, expected)


def test_range_aggregates_edge_cases(hip):
    for name, ts, values in cases.edge_case_series():
        batch = ora.try_compress_univariate_time_series(ts, values, cases.LOSSLESS)
        for t_lo, t_hi in ((int(ts[0]), int(ts[-1])), (int(ts[len(ts) // 2]), int(ts[-1])),
                           (int(ts[0]) - 5, int(ts[len(ts) // 2]))):
            _assert_state(hip.agg_batch_range(batch, t_lo, t_hi, ALL),
                          ora.agg_batch_range(batch, t_lo, t_hi, ALL))


def test_sums_of_irregular_swing_segments_are_the_same_either_way(hip, monkeypatch):
    # The walk adds up (slope * t + intercept) over a Swing segment's timestamps in the order of the points, as
    # the lane that decodes the stream by itself does: the two ways agree to the last bit, per segment.
    rng = np.random.default_rng(181)
    n = 120_000
    for timestamps in (1_600_000_000_000_000 + np.cumsum(rng.integers(900, 1100, n).astype(np.int64)),
                       1_600_000_000_000_000 + np.cumsum(np.where(rng.random(n) < 0.01, 2000, 1000).astype(np.int64))):
        values = (np.linspace(-5.0, 9.0, n) + 2 * np.sin(np.arange(n) / 900.0)).astype(np.float32)
        offsets = np.arange(0, n + 1, 4000, dtype=np.uint64)
        segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()["rel5"])
        assert (segments.model_type_id == mdb.MDB_SWING_ID).sum() > 10
        for row in range(0, len(segments), 7):
            one = segments.take(np.array([row]))
            monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)
            walked = hip.agg_batch(one, ALL)
            monkeypatch.setenv("MDB_AGG_TS_WALK", "0")
            alone = hip.agg_batch(one, ALL)
            assert (walked.count, walked.min, walked.max) == (alone.count, alone.min, alone.max)
            assert np.float64(walked.sum).tobytes() == np.float64(alone.sum).tobytes()
        # ... and under a time range: the points inside it, as GridExec + filter + aggregate see them
        for t_lo, t_hi in ((int(timestamps[n // 4]), int(timestamps[3 * n // 4])), (int(timestamps[5]) + 1, int(timestamps[4100])),
                           (int(timestamps[-3]), int(timestamps[-1]) + 10), (int(timestamps[0]) - 5, int(timestamps[0]))):
            monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)
            walked = hip.agg_batch_range(segments, t_lo, t_hi, ALL)
            monkeypatch.setenv("MDB_AGG_TS_WALK", "0")
            alone = hip.agg_batch_range(segments, t_lo, t_hi, ALL)
            assert (walked.count, walked.min, walked.max) == (alone.count, alone.min, alone.max), (t_lo, t_hi)
            assert np.float64(walked.sum).tobytes() == np.float64(alone.sum).tobytes(), (t_lo, t_hi)
            expected = ora.agg_batch_range(segments, t_lo, t_hi, ALL)
            assert (walked.count, walked.min, walked.max) == (expected.count, expected.min, expected.max)
        monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)
        _assert_state(hip.agg_batch(segments, ALL), ora.agg_batch(segments, ALL))


def test_sum_over_resident_batches_comes_from_the_cursor_index(hip):
    # A batch that stays on the device: SUM builds (or finds) the cursors into its MacaqueV streams, decodes every
    # piece of 64 values with a lane of its own and adds every stream up in stream order with one lane
    # (k_agg_mv_pieces, k_agg_mv_chains) - the f32 sums macaque_v.rs:220-265 produces, whoever decodes.
    import datagen
    n = 1_000_000
    timestamps, values = datagen.sine_series(3, n)
    offsets = np.arange(0, n + 65536, 65536, dtype=np.uint64)
    offsets[-1] = n
    segments = ora.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    resident = hip.upload_segments(segments)
    hip.profile_enable(True)
    hip.profile_reset()
    state = hip.agg_batch_dev(resident, ALL)          # (no grid call before it: the aggregate call builds the index)
    kernels = hip.profile()
    hip.profile_enable(False)
    assert "k_agg_mv_chains" in kernels and "k_mv_index_walk" in kernels and "k_mv_serial_sums" not in kernels
    _assert_state(state, ora.agg_batch(segments, ALL))
    per_stream = 0.0
    for k in range(len(offsets) - 1):
        per_stream += float(np.add.accumulate(values[int(offsets[k]):int(offsets[k + 1])], dtype=np.float32)[-1])
    assert abs(state.sum - per_stream) <= 1e-12 * abs(per_stream)
    resident.free()
    # residual tails with the seeds sum() uses (the model's last DECODED value), streams of every length, specials
    rng = np.random.default_rng(83)
    batches = [cases.edge_case_batch(), cases.edge_case_batch(cases.error_bounds()["rel5"])]
    for eb_name in ("lossless", "abs0.01", "rel1", "rel5"):
        for irregular in (False, True):
            batches.append(cases.mixed_batch(cases.error_bounds()[eb_name], irregular, seed=int(rng.integers(1, 1000)),
                                             length=25_000)[2])
    for batch in batches:
        resident = hip.upload_segments(batch)
        expected = ora.agg_batch(batch, ALL)
        for _ in range(2):
            _assert_state(hip.agg_batch_dev(resident, ALL), expected)
        lo, hi = int(batch.start_time[len(batch) // 3]), int(batch.end_time[2 * len(batch) // 3])
        _assert_state(hip.agg_batch_range_dev(resident, lo, hi, ALL), ora.agg_batch_range(batch, lo, hi, ALL))
        resident.free()


def test_a_resident_batch_keeps_what_the_walk_of_its_timestamps_found(hip, monkeypatch):
    # Aggregates without a time range over segments with irregular timestamps: len() and swing::sum come from a walk
    # of the streams that is the same every time, so a batch that stays on the device keeps its result - the counts
    # from the first call on, the sums from the first call that asks for them - and later calls do not walk.
    monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)   # (the mode without the walk has nothing to keep)
    rng = np.random.default_rng(29)
    n = 200_000
    timestamps = 1_600_000_000_000_000 + np.cumsum(rng.integers(900, 1100, n).astype(np.int64))
    values = (np.linspace(-5.0, 9.0, n) + 2 * np.sin(np.arange(n) / 900.0)).astype(np.float32)
    offsets = np.arange(0, n + 1, 4000, dtype=np.uint64)
    segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()["rel5"])
    expected = ora.agg_batch(segments, ALL)
    transient = hip.agg_batch(segments, ALL)
    resident = hip.upload_segments(segments)

    def call(mask):
        hip.profile_enable(True)
        hip.profile_reset()
        state = hip.agg_batch_dev(resident, mask)
        kernels = hip.profile()
        hip.profile_enable(False)
        return state, kernels

    counted, kernels = call(mdb.MDB_AGG_COUNT)
    assert "k_grid_ts_count" in kernels and counted.count == n
    counted, kernels = call(mdb.MDB_AGG_COUNT)
    assert "k_grid_ts_count" not in kernels and counted.count == n
    state, kernels = call(ALL)                      # the sums have not been asked for yet: one more walk
    assert "k_grid_ts_count" in kernels
    for _ in range(2):
        again, kernels = call(ALL)
        assert "k_grid_ts_count" not in kernels
        for got in (state, again):
            assert (got.count, got.min, got.max) == (transient.count, transient.min, transient.max)
            assert np.float64(got.sum).tobytes() == np.float64(transient.sum).tobytes()
    _assert_state(state, expected)
    monkeypatch.setenv("MDB_GRID_TS_CACHE", "0")    # (the switch: walk every time)
    again, kernels = call(ALL)
    assert "k_grid_ts_count" in kernels
    assert np.float64(again.sum).tobytes() == np.float64(transient.sum).tobytes()
    resident.free()


def test_a_resident_batch_walks_only_the_segments_a_time_range_cuts(hip, monkeypatch):
    # WHERE timestamp BETWEEN over segments with irregular timestamps: a batch that stays on the device keeps what a
    # walk over the whole time axis found, a query takes that for the segments its range contains and walks the ones
    # it cuts - and must say what the walk of every segment (a batch on the host, or MDB_GRID_TS_CACHE=0) says.
    monkeypatch.delenv("MDB_AGG_TS_WALK", raising=False)
    monkeypatch.delenv("MDB_GRID_TS_CACHE", raising=False)
    rng = np.random.default_rng(31)
    series, points = 6, 40_000
    one = 1_600_000_000_000_000 + np.cumsum(rng.integers(900, 1100, points).astype(np.int64))
    timestamps = np.tile(one, series)
    values = np.concatenate([(np.linspace(-5.0, 9.0, points) * (k + 1) + 2 * np.sin(np.arange(points) / (300.0 + 90 * k))
                              + (rng.normal(0, 0.3, points) if k % 3 == 2 else 0)).astype(np.float32) for k in range(series)])
    offsets = np.array([k * points + c for k in range(series) for c in range(0, points, 4000)] + [series * points], dtype=np.uint64)
    for eb_name in ("rel5", "lossless"):
        segments = hip.compress_chunks(timestamps, values, offsets, cases.error_bounds()[eb_name])
        resident = hip.upload_segments(segments)
        ranges = [(int(one[0]), int(one[-1])), (int(one[0]) - 5, int(one[-1]) + 5), (int(one[123]), int(one[123])),
                  (int(one[-1]) + 1, int(one[-1]) + 100), (int(one[0]) - 100, int(one[0]) - 1)]
        for _ in range(6):
            a, b = sorted(int(x) for x in rng.integers(0, points, 2))
            ranges.append((int(one[a]) + int(rng.integers(-2, 3)), int(one[b]) + int(rng.integers(-2, 3))))
        first = True
        for t_lo, t_hi in ranges:
            if t_lo > t_hi:
                continue
            hip.profile_enable(True)
            hip.profile_reset()
            got = hip.agg_batch_range_dev(resident, t_lo, t_hi, ALL)
            kernels = hip.profile()
            hip.profile_enable(False)
            assert "k_ts_range_select" in kernels
            if not first and (t_lo, t_hi) == ranges[0]:
                assert "k_grid_ts_count" not in kernels
            first = False
            transient = hip.agg_batch_range(segments, t_lo, t_hi, ALL)
            assert (got.count, got.min, got.max) == (transient.count, transient.min, transient.max), (eb_name, t_lo, t_hi)
            assert np.float64(got.sum).tobytes() == np.float64(transient.sum).tobytes(), (eb_name, t_lo, t_hi)
            _assert_state(got, ora.agg_batch_range(segments, t_lo, t_hi, ALL))
        # (the whole range a second time: nothing is cut, nothing is walked)
        hip.profile_enable(True)
        hip.profile_reset()
        hip.agg_batch_range_dev(resident, *ranges[0], ALL)
        assert "k_grid_ts_count" not in hip.profile()
        hip.profile_enable(False)
        monkeypatch.setenv("MDB_GRID_TS_CACHE", "0")
        hip.profile_enable(True)
        hip.profile_reset()
        again = hip.agg_batch_range_dev(resident, *ranges[-1], ALL)
        kernels = hip.profile()
        hip.profile_enable(False)
        assert "k_ts_range_select" not in kernels
        transient = hip.agg_batch_range(segments, *ranges[-1], ALL)
        assert np.float64(again.sum).tobytes() == np.float64(transient.sum).tobytes()
        monkeypatch.delenv("MDB_GRID_TS_CACHE")
        resident.free()


def test_range_aggregates_over_long_lossless_streams_go_piece_by_piece(hip, monkeypatch):
    # WHERE timestamp BETWEEN over MacaqueV segments with cursors into their streams (a resident batch's sidecar, or the
    # ones the call's host threads leave): only the pieces of 64 values that reach into the range are decoded
    # (k_agg_mv_range). COUNT / MIN / MAX as GridExec + filter + aggregate, SUM to its tolerance; and the same as without
    # the cursors (MDB_AGG_RANGE_PIECES=0).
    import datagen
    monkeypatch.delenv("MDB_AGG_RANGE_PIECES", raising=False)
    n = 400_000
    timestamps, values = datagen.sine_series(5, n)
    offsets = np.arange(0, n + 50_000, 50_000, dtype=np.uint64)
    offsets[-1] = n
    segments = ora.compress_chunks(timestamps, values, offsets, cases.LOSSLESS)
    assert (segments.model_type_id == mdb.MDB_MACAQUE_V_ID).sum() >= 8
    resident = hip.upload_segments(segments)
    ranges = [(int(timestamps[n // 4]), int(timestamps[3 * n // 4])),          # whole segments and two cut ones
              (int(timestamps[70_001]), int(timestamps[70_130])),              # inside three pieces of one stream
              (int(timestamps[99_990]) + 1, int(timestamps[100_010]) - 1),      # across two segments
              (int(timestamps[0]) - 10, int(timestamps[0])),                   # the first point alone
              (int(timestamps[-1]) + 1, int(timestamps[-1]) + 100),            # behind everything
              (int(timestamps[0]) - 10, int(timestamps[-1]) + 10)]             # everything
    for t_lo, t_hi in ranges:
        expected = ora.agg_batch_range(segments, t_lo, t_hi, ALL)
        inside = (timestamps >= t_lo) & (timestamps <= t_hi)
        assert expected.count == int(inside.sum())
        for where in ("resident", "host"):
            call = ((lambda: hip.agg_batch_range_dev(resident, t_lo, t_hi, ALL)) if where == "resident"
                    else (lambda: hip.agg_batch_range(segments, t_lo, t_hi, ALL)))
            hip.profile_enable(True)
            hip.profile_reset()
            by_pieces = call()
            kernels = hip.profile()
            hip.profile_enable(False)
            if where == "resident" or int(inside.sum()) > 0:
                assert "k_agg_mv_range" in kernels, (where, t_lo, t_hi, sorted(kernels))
            _assert_state(by_pieces, expected)
            if expected.count:
                assert np.float32(by_pieces.min) == values[inside].min() and np.float32(by_pieces.max) == values[inside].max()
            monkeypatch.setenv("MDB_AGG_RANGE_PIECES", "0")
            hip.profile_enable(True)
            hip.profile_reset()
            without = call()
            kernels = hip.profile()
            hip.profile_enable(False)
            monkeypatch.delenv("MDB_AGG_RANGE_PIECES")
            assert "k_agg_mv_range" not in kernels
            assert (without.count, without.min, without.max) == (by_pieces.count, by_pieces.min, by_pieces.max)
            assert abs(without.sum - by_pieces.sum) <= 1e-9 * max(abs(by_pieces.sum), 1e-30)
    resident.free()


def test_macaque_streams_that_reach_beyond_a_wave_are_added_up_in_stream_order(hip):
    # SUM over a resident batch of many segments with long MacaqueV streams among them (lossless noise: a chunk is one
    # stream) and short ones cut by the ends of the piece kernel's waves: k_agg_mv_chain_list lists the streams that
    # k_agg_mv_pieces could not sum inside one wave, k_agg_mv_chain_groups adds each of them up with eight lanes loading
    # and one adding - value after value, as macaque_v::sum does (macaque_v.rs:228-235): every segment's f32 sum is
    # numpy's sequential f32 sum of its values bit for bit, the f64 total the oracle's, and the same on every call.
    rng = np.random.default_rng(47)
    parts, offsets = [], [0]
    for length in (70_000, 4_096, 4_095, 12_345, 5_000, 65_536):   # noise: one MacaqueV segment per chunk
        parts.append(rng.uniform(-1e3, 1e3, length).astype(np.float32))
        offsets.append(offsets[-1] + length)
    for _ in range(40):                                            # ... and chunks of many short segments
        runs = [np.full(int(rng.integers(9, 40)), float(rng.uniform(-50, 50)), dtype=np.float32) for _ in range(60)]
        runs += [rng.uniform(-1e3, 1e3, int(rng.integers(1, 300))).astype(np.float32) for _ in range(20)]
        order = rng.permutation(len(runs))
        parts.append(np.concatenate([runs[k] for k in order]))
        offsets.append(offsets[-1] + len(parts[-1]))
    values = np.concatenate(parts)
    timestamps = 1_700_000_000_000 + 100 * np.arange(len(values), dtype=np.int64)
    segments = hip.compress_chunks(timestamps, values, np.array(offsets, dtype=np.uint64), cases.LOSSLESS)
    assert len(segments) > 512 and int((segments.model_type_id == 2).sum()) >= 6
    resident = hip.upload_segments(segments)
    hip.profile_enable(True)
    hip.profile_reset()
    state = hip.agg_batch_dev(resident, ALL)
    kernels = hip.profile()
    hip.profile_enable(False)
    assert "k_agg_mv_chains" in kernels
    _assert_state(state, ora.agg_batch(segments, ALL))
    # every MacaqueV segment's values added up one after the other in f32 (the first one IS the sum's start), the
    # segments' sums in f64: what the reference's accumulator makes of them
    total = 0.0
    for row in np.nonzero(segments.model_type_id == 2)[0]:
        first = int(np.searchsorted(timestamps, segments.start_time[row]))
        last = int(np.searchsorted(timestamps, segments.end_time[row]))
        total += float(np.cumsum(values[first:last + 1], dtype=np.float32)[-1])
    others = hip.agg_batch(segments.take(np.nonzero(segments.model_type_id != 2)[0]), ALL)
    assert abs(state.sum - (total + others.sum)) <= 1e-12 * abs(state.sum)
    # Where the waves of pieces list their streams is a function of the cursors: counted by the first call, kept with
    # the batch's index for the later ones (MDB_AGG_KEEP_CHAIN_OFFSETS=0: counted by every call) - the same sums either way.
    assert "k_agg_mv_chain_count" in kernels
    hip.profile_enable(True)
    hip.profile_reset()
    again = hip.agg_batch_dev(resident, ALL)
    later = hip.profile()
    assert "k_agg_mv_chain_count" not in later and "k_agg_mv_chains" in later and "k_agg_mv_pieces" in later
    assert np.float64(again.sum).tobytes() == np.float64(state.sum).tobytes() and again.count == state.count
    os.environ["MDB_AGG_KEEP_CHAIN_OFFSETS"] = "0"
    try:
        hip.profile_reset()
        counted = hip.agg_batch_dev(resident, ALL)
        assert "k_agg_mv_chain_count" in hip.profile()
    finally:
        del os.environ["MDB_AGG_KEEP_CHAIN_OFFSETS"]
        hip.profile_enable(False)
    assert np.float64(counted.sum).tobytes() == np.float64(state.sum).tobytes() and counted.count == state.count
    resident.free()
