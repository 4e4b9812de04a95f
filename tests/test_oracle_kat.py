"""Pins the CPU oracle against the known-answer tests of the reference's own unit tests.

Every test names the reference test it restates (file:line relative to
crates/modelardb_compression/src/ unless stated otherwise). The reference cannot be compiled in the
authoring container (no Rust toolchain), so these vectors are what anchors the oracle.
"""

import struct

import numpy as np
import pytest

import datagen
import oracle_lib as ora
from modelardb_rs_amd import (MDB_MACAQUE_V_ID, MDB_PMC_MEAN_ID, MDB_SWING_ID, error_bound)

LOSSLESS = error_bound("lossless")
ABS_FIVE = error_bound("absolute", 5.0)
REL_FIVE = error_bound("relative", 5.0)
ABS_TEN = error_bound("absolute", 10.0)
REL_TEN = error_bound("relative", 10.0)
ABS_ONE = error_bound("absolute", 1.0)
ABS_MAX = error_bound("absolute", float(np.finfo(np.float32).max))
REL_MAX = error_bound("relative", 100.0)
F32_MIN = -float(np.finfo(np.float32).max)  # Rust's f32::MIN
F32_MAX = float(np.finfo(np.float32).max)

SPECIAL_F32 = [0.0, -0.0, 1.0, -1.0, 37.0, 1e-45, -1e-45, 1.1754942e-38, 3.4028235e38,
               -3.4028235e38, float("inf"), float("-inf"), float("nan"), 0.1, 123456.79]


# ---- models/bits.rs ---------------------------------------------------------------------------

TEST_BYTES = bytes([255, 170, 0])
TEST_BITS = [1] * 8 + [1, 0, 1, 0, 1, 0, 1, 0] + [0] * 8


def test_reading_the_test_bits():  # bits.rs:188-208
    values, remaining = ora.bits_read(TEST_BYTES, [1] * 24)
    assert values == TEST_BITS
    assert remaining == 0


def test_bit_reader_cannot_be_empty():  # bits.rs:195-203
    with pytest.raises(ora.OracleError, match="The bytes slice must not be empty."):
        ora.bits_read(b"", [1])


def test_remaining_bits():  # bits.rs:211-221
    for widths, expected in (([4], 12), ([4, 8], 4), ([4, 8, 4], 0)):
        _, remaining = ora.bits_read(bytes([0, 255]), widths)
        assert remaining == expected


def test_writing_and_reading_the_test_bits():  # bits.rs:302-312
    assert ora.bits_write([(b, 1) for b in TEST_BITS]) == TEST_BYTES


def test_bit_vec_builder_lengths():  # bits.rs:224-274
    assert ora.bits_write([]) == b""
    assert len(ora.bits_write([(1, 1)])) == 1
    assert len(ora.bits_write([(0, 1)])) == 1
    assert len(ora.bits_write([(1, 1)] * 8)) == 1
    assert len(ora.bits_write([(1, 1)] * 9)) == 2
    assert len(ora.bits_write([(0, 1)] * 9)) == 2


def test_finish_with_one_bits():  # bits.rs:276-299
    assert ora.bits_write([], finish_with_ones=True) == b""
    assert ora.bits_write([(255, 8)], finish_with_ones=True) == bytes([255])
    assert ora.bits_write([(15, 4)], finish_with_ones=True) == bytes([255])


def test_writing_and_reading_random_bits():  # bits.rs:314-324 (proptest restated with numpy)
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(1, 50))
        bits = [int(b) for b in rng.integers(0, 2, size=n)]
        data = ora.bits_write([(b, 1) for b in bits])
        assert ora.bits_read(data, [1] * n)[0] == bits


def test_multi_bit_fields_round_trip():
    rng = np.random.default_rng(2)
    for _ in range(200):
        widths = [int(w) for w in rng.integers(1, 65, size=int(rng.integers(1, 20)))]
        fields = [int(rng.integers(0, 1 << 62)) & ((1 << w) - 1) for w in widths]
        data = ora.bits_write(list(zip(fields, widths)))
        assert ora.bits_read(data, widths)[0] == fields


# ---- models/timestamps.rs ---------------------------------------------------------------------

def _ts_round_trip(timestamps, known_size=None):  # timestamps.rs:428-454
    compressed = ora.compress_residual_timestamps(timestamps)
    assert (len(timestamps) <= 2) == (len(compressed) == 0)
    if known_size is not None:
        assert len(compressed) == known_size
    out = ora.decompress_all_timestamps(timestamps[0], timestamps[-1], compressed)
    assert out.tolist() == list(timestamps)
    return compressed


def test_compress_timestamps_zero_one_or_two():  # timestamps.rs:304-318
    assert ora.compress_residual_timestamps([]) == b""
    assert ora.compress_residual_timestamps([100]) == b""
    assert ora.compress_residual_timestamps([100, 300]) == b""


def test_regular_time_series():  # timestamps.rs:321-332
    data = _ts_round_trip([1579701905500 + 100 * i for i in range(5)], 1)
    assert data == bytes([5])


def test_irregular_time_series():  # timestamps.rs:335-346
    _ts_round_trip([1579694400057, 1579694400197, 1579694400353, 1579694400493, 1579694400650], 4)


@pytest.mark.parametrize("timestamps,size", [
    ([100, 100, 200], 1),                                      # :349-357 bucket 0
    ([100, 37, 38, 200], 3),                                   # :360-369 7 bits, -63 and 64
    ([500, 245, 246, 500], 4),                                 # :372-381 9 bits, -255 and 256
    ([5000, 2953, 2954, 5000], 5),                             # :384-393 12 bits, -2047 and 2048
    ([5000000000, 2852516353, 2852516354, 5000000000], 10),    # :396-405 32 bits
])
def test_bucket_sizes(timestamps, size):
    _ts_round_trip(timestamps, size)


def test_64_bit_bucket():
    _ts_round_trip([0, 1 << 40, (1 << 40) + 5, 1 << 41])


def test_generated_time_series_round_trip():  # timestamps.rs:408-417
    _ts_round_trip(datagen.generate_timestamps(1000, False).tolist())
    _ts_round_trip(datagen.generate_timestamps(1000, True, np.random.default_rng(3)).tolist())


def test_random_irregular_time_series_round_trip():  # timestamps.rs:419-426
    rng = np.random.default_rng(4)
    for _ in range(100):
        n = int(rng.integers(1, 50))
        ts = np.sort(np.abs(rng.integers(-(1 << 62), 1 << 62, size=n)))
        _ts_round_trip([int(t) for t in ts])


def test_derived_regular_length_encodings():  # derived from timestamps.rs:99-108
    expected = {3: "03", 5: "05", 127: "7f", 128: "0080", 255: "00ff", 65536: "010000"}
    for n, encoded in expected.items():
        data = ora.compress_residual_timestamps(np.arange(n, dtype=np.int64) * 10)
        assert data.hex() == encoded
        assert ora.are_compressed_timestamps_regular(data)
        assert ora.seg_len(0, (n - 1) * 10, data) == n


def test_regularity_of_compressed_timestamps():  # timestamps.rs:199-202
    assert ora.are_compressed_timestamps_regular(b"")
    assert ora.are_compressed_timestamps_regular(bytes([0x7F]))
    assert not ora.are_compressed_timestamps_regular(bytes([0x80]))
    irregular = ora.compress_residual_timestamps([100, 150, 300, 350, 700, 750, 1500])  # :473-478
    assert not ora.are_compressed_timestamps_regular(irregular)
    regular = ora.compress_residual_timestamps([100, 200, 300, 400, 500, 600, 700])  # :466-470
    assert ora.are_compressed_timestamps_regular(regular)


# ---- models/mod.rs ------------------------------------------------------------------------------

def test_len_of_segments():  # mod.rs:409-416
    assert ora.seg_len(1658671178037, 1658671178037, b"") == 1
    assert ora.seg_len(1658671178037, 1658671187047, bytes([10])) == 10
    assert ora.seg_len(100, 200, b"") == 2


def test_decompress_and_split_into_models_and_residuals():  # mod.rs:434-465
    ts, values = ora.seg_grid(MDB_PMC_MEAN_ID, 100, 500, bytes([5]), 1.0, 1.0, b"", b"")
    assert ts.tolist() == [100, 200, 300, 400, 500]
    assert values.tolist() == [1.0] * 5
    # residuals = two repeated values (`10` `10`) + count byte 2 -> model [100,200,300].
    residuals = ora.macaque_v_compress(LOSSLESS, [1.0, 1.0], seed=1.0)[0] + bytes([2])
    ts, values = ora.seg_grid(MDB_PMC_MEAN_ID, 100, 500, bytes([5]), 1.0, 1.0, b"", residuals)
    assert ts.tolist() == [100, 200, 300, 400, 500]
    assert values.tolist() == [1.0] * 5


def test_error_bound_known_answers():  # mod.rs:390-405
    assert ora.is_value_within_error_bound(ABS_ONE, 10.0, 11.0)
    assert ora.is_value_within_error_bound(REL_TEN, 10.0, 11.0)
    assert not ora.is_value_within_error_bound(LOSSLESS, 10.0, 11.0)


def test_error_bound_special_values():  # mod.rs:298-387 (proptests restated over special values)
    inf, nan = float("inf"), float("nan")
    for value in SPECIAL_F32:
        assert ora.is_value_within_error_bound(LOSSLESS, value, value)
        for eb in (ABS_MAX, REL_MAX):
            for special in (inf, -inf):
                if value != special:
                    assert not ora.is_value_within_error_bound(eb, special, value)
                    assert not ora.is_value_within_error_bound(eb, value, special)
            if not np.isnan(value):
                assert not ora.is_value_within_error_bound(eb, nan, value)
                assert not ora.is_value_within_error_bound(eb, value, nan)


def test_maximum_allowed_deviation():  # mod.rs:83-90
    assert ora.maximum_allowed_deviation(LOSSLESS, 123.0) == 0.0
    assert ora.maximum_allowed_deviation(ABS_FIVE, 123.0) == 5.0 * 0.99
    assert ora.maximum_allowed_deviation(REL_FIVE, -200.0) == abs(-200.0 * (5.0 / 100.1))


# ---- models/pmc_mean.rs -------------------------------------------------------------------------

def test_pmc_fits_repeated_values_lossless():  # pmc_mean.rs:141-152 (incl. NaN / inf)
    for value in SPECIAL_F32:
        n_fit, model, _ = ora.pmc_mean_fit(LOSSLESS, [value] * 5)
        assert n_fit == 5
        assert (np.isnan(model) and np.isnan(value)) or model == np.float32(value)


def test_pmc_known_sequences():  # pmc_mean.rs:269-299
    values = [42.0, 42.0, 42.8, 42.0, 42.0]
    assert ora.pmc_mean_fit(LOSSLESS, values)[0] < 5
    assert ora.pmc_mean_fit(ABS_FIVE, values)[0] == 5
    assert ora.pmc_mean_fit(REL_FIVE, values)[0] == 5


def test_pmc_other_value_never_fits_non_finite():  # pmc_mean.rs:154-267 (restated)
    for special in (float("inf"), float("-inf"), float("nan")):
        for eb in (ABS_MAX, REL_MAX):
            assert ora.pmc_mean_fit(eb, [special, 1.0])[0] == 1
            assert ora.pmc_mean_fit(eb, [1.0, special])[0] == 1


def test_pmc_sum():  # pmc_mean.rs:304-306
    for value in (0.5, 37.0, -3.25):
        total = ora.seg_sum(MDB_PMC_MEAN_ID, 0, 900, bytes([10]), value, value, b"", b"")
        assert total == np.float32(10) * np.float32(value)


def test_pmc_bytes_per_value():  # pmc_mean.rs:83-87 with the derived constant 29
    assert ora.pmc_mean_fit(LOSSLESS, [1.0] * 8)[2] == np.float32(29.0) / np.float32(8.0)


# ---- models/swing.rs ----------------------------------------------------------------------------

START_TIME = 1658671178037
INTERVAL = 1000


def _swing_ts(n):
    return [START_TIME + INTERVAL * i for i in range(n)]


def test_swing_known_sequences():  # swing.rs:539-569
    linear = [42.0, 84.0, 126.0, 168.0, 210.0]
    different = [42.0, 42.0, 42.8, 42.0, 42.0]
    assert ora.swing_fit(LOSSLESS, _swing_ts(5), linear)[0] == 5
    assert ora.swing_fit(LOSSLESS, _swing_ts(5), different)[0] < 5
    assert ora.swing_fit(ABS_FIVE, _swing_ts(5), different)[0] == 5
    assert ora.swing_fit(REL_FIVE, _swing_ts(5), different)[0] == 5


def test_swing_non_finite_values():  # swing.rs:366-537 (restated)
    for special in (float("inf"), float("-inf"), float("nan")):
        assert ora.swing_fit(LOSSLESS, _swing_ts(5), [special] * 5)[0] == 5
        for eb in (ABS_MAX, REL_MAX):
            assert ora.swing_fit(eb, _swing_ts(2), [special, 1.0])[0] == 1
            assert ora.swing_fit(eb, _swing_ts(2), [1.0, special])[0] == 1


SLOPE_VALUES = [42.0, 42.0, 42.8, 42.0, 41.0, 40.0, 42.0, 42.0, 42.0, 42.1]


def test_swing_slope_is_between_hyperplanes():  # swing.rs:571-577, 611-633
    n_fit, first, last, _, bounds = ora.swing_fit(REL_FIVE, _swing_ts(10), SLOPE_VALUES)
    assert n_fit == 10
    end_time = START_TIME + 10 * INTERVAL  # the reference test uses one interval past the end
    slope = (float(last) - float(first)) / float(end_time - START_TIME)
    assert bounds[2] <= slope <= bounds[0]


def test_swing_can_minimize_mse():  # swing.rs:579-609
    timestamps = _swing_ts(10)
    n_fit, first, last, _, _ = ora.swing_fit(REL_FIVE, timestamps, SLOPE_VALUES)
    assert n_fit == 10

    def mse(v_first, v_last):
        slope = (v_last - v_first) / (timestamps[-1] - timestamps[0])
        return sum((v_first + slope * (t - timestamps[0]) - v) ** 2
                   for t, v in zip(timestamps, SLOPE_VALUES)) / 10

    assert mse(SLOPE_VALUES[0], SLOPE_VALUES[-1]) > mse(float(first), float(last))


def test_swing_sum():  # swing.rs:668-677
    rng = np.random.default_rng(5)
    for _ in range(200):
        a = float(int(rng.integers(-(1 << 31), 1 << 31)) % 1_000_000)
        b = float(int(rng.integers(-(1 << 31), 1 << 31)) % 1_000_000)
        assert ora.swing_sum(START_TIME, START_TIME + INTERVAL, b"", a, b, 0) == np.float32(a + b)


def test_swing_grid_constant():  # swing.rs:679-706
    for value in (-999999.0, 0.0, 1.0, 424242.0):
        values_column = bytes([0])  # first !< last -> decreasing flag (types.rs:128-132)
        ts, values = ora.seg_grid(MDB_SWING_ID, START_TIME, START_TIME + INTERVAL, b"", value,
                                  value, values_column, b"")
        assert ts.tolist() == [START_TIME, START_TIME + INTERVAL]
        assert values.tolist() == [value, value]


@pytest.mark.parametrize("decreasing", [False, True])
def test_swing_reconstructs_linear_sequence(decreasing):  # swing.rs:717-798
    values = np.arange(42, 4201, 42, dtype=np.float32)
    if decreasing:
        values = values[::-1].copy()
    timestamps = np.array(_swing_ts(len(values)), dtype=np.int64)
    batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
    assert len(batch) == 1
    assert batch.model_type_id[0] == MDB_SWING_ID
    ts, reconstructed, _, _ = ora.grid_batch(batch)
    assert np.array_equal(ts, timestamps)
    assert np.array_equal(reconstructed, values)  # bit equal


# ---- models/macaque_v.rs ------------------------------------------------------------------------

def test_macaque_v_empty():  # macaque_v.rs:349-352
    assert ora.macaque_v_compress(LOSSLESS, [])[0] == b""


def test_macaque_v_single_and_repeated_values():  # macaque_v.rs:354-376
    for value in SPECIAL_F32:
        for values in ([value], [value, value]):
            _, _, _, lz, tz, last = ora.macaque_v_compress(LOSSLESS, values)
            assert (np.isnan(last) and np.isnan(value)) or last == np.float32(value)
            assert (lz, tz) == (255, 0)


def test_macaque_v_leading_and_trailing_zero_bits():  # macaque_v.rs:378-398
    for values in ([37.0, 73.0], [37.0, 71.0, 73.0]):
        _, _, _, lz, tz, last = ora.macaque_v_compress(LOSSLESS, values)
        assert (lz, tz, last) == (8, 17, 73.0)


@pytest.mark.parametrize("eb", [ABS_TEN, REL_TEN])
def test_macaque_v_value_within_error_bound_leaves_state(eb):  # macaque_v.rs:400-433
    _, _, _, lz0, tz0, last0 = ora.macaque_v_compress(eb, [10.0])
    _, _, _, lz1, tz1, last1 = ora.macaque_v_compress(eb, [10.0, 11.0])
    assert (lz0, tz0, last0) == (lz1, tz1, last1)


def test_macaque_v_sum_and_grid_lossless():  # macaque_v.rs:436-475
    rng = np.random.default_rng(6)
    pool = np.array(SPECIAL_F32, dtype=np.float32)
    for _ in range(300):
        n = int(rng.integers(1, 50))
        values = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32).view(np.float32)
        mask = rng.random(n) < 0.2
        values = np.where(mask, pool[rng.integers(0, len(pool), size=n)], values).astype(np.float32)
        data = ora.macaque_v_compress(LOSSLESS, values)[0]
        decoded = ora.macaque_v_grid(data, n)
        assert np.array_equal(decoded.view(np.uint32), values.view(np.uint32))
        expected = np.float32(values[0])
        with np.errstate(all="ignore"):
            for v in values[1:]:
                expected = np.float32(expected + v)
        total = ora.macaque_v_sum(data, n)
        assert (np.isnan(total) and np.isnan(expected)) or total == expected


def test_macaque_v_single_value_with_and_without_seed():  # macaque_v.rs:449-464, 495-521
    for seed in (None, 37.0):
        data = ora.macaque_v_compress(LOSSLESS, [37.0], seed=seed)[0]
        assert ora.macaque_v_sum(data, 1, seed=seed) == 37.0
        assert ora.macaque_v_grid(data, 1, seed=seed).tolist() == [37.0]


def test_macaque_v_derived_bytes():  # derived from macaque_v.rs:76-164, length pinned
    data, mn, mx, _, _, _ = ora.macaque_v_compress(LOSSLESS, [73.0, 37.0, 37.0, 37.0, 73.0])
    assert data.hex() == "42920000d03c3a43"
    assert (mn, mx) == (37.0, 73.0)


def test_macaque_v_lossy_values_stay_within_bound():
    rng = np.random.default_rng(7)
    for eb in (ABS_FIVE, REL_FIVE, REL_TEN, error_bound("relative", 0.1),
               error_bound("absolute", 0.01)):
        values = (100 + 50 * rng.standard_normal(500)).astype(np.float32)
        data = ora.macaque_v_compress(eb, values)[0]
        decoded = ora.macaque_v_grid(data, len(values))
        for real, approximate in zip(values, decoded):
            assert ora.is_value_within_error_bound(eb, float(real), float(approximate))


# ---- types.rs -----------------------------------------------------------------------------------

UNCOMPRESSED_TIMESTAMPS = [100, 200, 300, 400, 500]


def _create_segment(values, model_type_id, end_index, model_min, model_max, model_values_len,
                    segment_min, segment_max, segment_values_len):  # types.rs:791-860
    model = ora.fit_next_model(0, LOSSLESS, UNCOMPRESSED_TIMESTAMPS, values)
    assert model.model_type_id == model_type_id
    assert model.start_index == 0
    assert model.end_index == end_index
    assert model.min_value == model_min
    assert model.max_value == model_max
    assert model.values_len == model_values_len
    batch = ora.model_finish(model, LOSSLESS, 4, UNCOMPRESSED_TIMESTAMPS, values)
    assert len(batch) == 1
    assert batch.min_value[0] == np.float32(segment_min)
    assert batch.max_value[0] == np.float32(segment_max)
    segment_values = batch.values.value(0)
    assert len(segment_values) == segment_values_len
    return model, batch, segment_values


@pytest.mark.parametrize("values,end_index,segment_min,segment_max,segment_values_len", [
    ([10.0, 10.0, 10.0, 10.0, 10.0], 4, 10.0, 10.0, 0),          # types.rs:536-547
    ([10.0, 10.0, 10.0, 10.0, F32_MIN], 3, F32_MIN, 10.0, 1),    # types.rs:550-561
    ([10.0, 10.0, 10.0, 10.0, F32_MAX], 3, 10.0, F32_MAX, 0),    # types.rs:564-575
    ([10.0, 10.0, 10.0, F32_MIN, F32_MAX], 2, F32_MIN, F32_MAX, 4),  # types.rs:578-589
])
def test_encoding_decoding_for_pmc_mean(values, end_index, segment_min, segment_max,
                                        segment_values_len):
    _, batch, segment_values = _create_segment(values, MDB_PMC_MEAN_ID, end_index, 10.0, 10.0, 0,
                                               segment_min, segment_max, segment_values_len)
    decoded = ora.decode_values_for_pmc_mean(batch.min_value[0], batch.max_value[0], segment_values)
    assert decoded == 10.0
    ts, reconstructed, _, _ = ora.grid_batch(batch)
    assert ts.tolist() == UNCOMPRESSED_TIMESTAMPS
    assert reconstructed.tolist() == [float(np.float32(v)) for v in values]


@pytest.mark.parametrize(
    "values,end_index,model_min,model_max,model_values_len,segment_min,segment_max,"
    "segment_values_len", [
        ([10.0, 20.0, 30.0, 40.0, 50.0], 4, 10.0, 50.0, 0, 10.0, 50.0, 0),          # :628-640
        ([10.0, 20.0, 30.0, 40.0, F32_MIN], 3, 10.0, 40.0, 0, F32_MIN, 40.0, 5),    # :643-655
        ([10.0, 20.0, 30.0, 40.0, F32_MAX], 3, 10.0, 40.0, 0, 10.0, F32_MAX, 5),    # :658-670
        ([10.0, 20.0, 30.0, F32_MIN, F32_MAX], 2, 10.0, 30.0, 0, F32_MIN, F32_MAX, 8),  # :673-685
        ([50.0, 40.0, 30.0, 20.0, 10.0], 4, 10.0, 50.0, 1, 10.0, 50.0, 1),          # :688-700
        ([50.0, 40.0, 30.0, 20.0, F32_MIN], 3, 20.0, 50.0, 1, F32_MIN, 50.0, 5),    # :703-715
        ([50.0, 40.0, 30.0, 20.0, F32_MAX], 3, 20.0, 50.0, 1, 20.0, F32_MAX, 5),    # :718-730
        ([50.0, 40.0, 30.0, F32_MIN, F32_MAX], 2, 30.0, 50.0, 1, F32_MIN, F32_MAX, 8),  # :733-745
    ])
def test_encoding_decoding_for_swing(values, end_index, model_min, model_max, model_values_len,
                                     segment_min, segment_max, segment_values_len):
    model, batch, segment_values = _create_segment(values, MDB_SWING_ID, end_index, model_min,
                                                   model_max, model_values_len, segment_min,
                                                   segment_max, segment_values_len)
    first, last = ora.decode_values_for_swing(batch.min_value[0], batch.max_value[0],
                                              segment_values)
    assert first == np.float32(values[0])           # types.rs:776-787
    assert last == np.float32(values[end_index])
    ts, reconstructed, _, _ = ora.grid_batch(batch)
    assert ts.tolist() == UNCOMPRESSED_TIMESTAMPS
    assert reconstructed.tolist() == [float(np.float32(v)) for v in values]


def test_model_with_fewest_bytes_is_selected():  # types.rs:862-890
    rng = np.random.default_rng(8)
    timestamps = datagen.generate_timestamps(25, False)
    constant = datagen.generate_values(timestamps, "constant", rng)
    random = datagen.generate_values(timestamps, "random", rng, value_range=(0.0, 100.0))
    values = np.concatenate([constant, random])
    timestamps = datagen.generate_timestamps(50, False)
    model = ora.fit_next_model(0, REL_TEN, timestamps, values)
    assert model.model_type_id == MDB_PMC_MEAN_ID


def test_values_column_codecs_direct():  # types.rs:283-407
    assert ora.encode_values_for_pmc_mean(10.0, 10.0, 10.0, 10.0) == b""
    assert ora.encode_values_for_pmc_mean(10.0, 10.0, 5.0, 10.0) == bytes([1])
    assert ora.encode_values_for_pmc_mean(10.0, 10.0, 5.0, 20.0) == struct.pack("<f", 10.0)
    assert ora.decode_values_for_pmc_mean(1.0, 2.0, b"") == 1.0
    assert ora.decode_values_for_pmc_mean(1.0, 2.0, bytes([1])) == 2.0
    assert ora.decode_values_for_pmc_mean(1.0, 2.0, struct.pack("<f", 7.5)) == 7.5
    with pytest.raises(ora.OracleError):
        ora.decode_values_for_pmc_mean(1.0, 2.0, b"ab")
    for increasing in (True, False):
        flag = lambda a, b: a if increasing else b
        assert ora.encode_values_for_swing(1.0, 9.0, increasing, 1.0, 9.0) == (
            b"" if increasing else bytes([0]))
        assert ora.encode_values_for_swing(1.0, 9.0, increasing, 0.0, 9.0) == (
            bytes([flag(0, 1)]) + struct.pack("<f", 1.0))
        assert ora.encode_values_for_swing(1.0, 9.0, increasing, 1.0, 10.0) == (
            bytes([flag(2, 3)]) + struct.pack("<f", 9.0))
        both = ora.encode_values_for_swing(1.0, 9.0, increasing, 0.0, 10.0)
        assert both == struct.pack("<ff", *((1.0, 9.0) if increasing else (9.0, 1.0)))
    assert ora.decode_values_for_swing(1.0, 9.0, b"") == (1.0, 9.0)
    assert ora.decode_values_for_swing(1.0, 9.0, bytes([0])) == (9.0, 1.0)
    for tag, expected in ((0, (5.0, 9.0)), (1, (9.0, 5.0)), (2, (1.0, 5.0)), (3, (5.0, 1.0))):
        data = bytes([tag]) + struct.pack("<f", 5.0)
        assert ora.decode_values_for_swing(1.0, 9.0, data) == expected
    assert ora.decode_values_for_swing(1.0, 9.0, struct.pack("<ff", 3.0, 4.0)) == (3.0, 4.0)
    with pytest.raises(ora.OracleError):
        ora.decode_values_for_swing(1.0, 9.0, b"abc")


# ---- compression.rs -----------------------------------------------------------------------------

def _assert_values_within(eb, values, reconstructed):
    """compression.rs:914-928, with one documented hazard of the reference itself: Swing tests its
    bounds in f64 (swing.rs:146-153) and never checks the f32 it finally stores, so once the f32
    spacing of a value approaches an ABSOLUTE bound (|v| >= 2^22 for a bound of 5) the rounded
    reconstruction can miss by up to bound + spacing. The reference's generator reaches 7.5e7 with
    the same recipe. Such misses are accepted only in that regime and only by that margin."""
    assert len(reconstructed) == len(values)
    for real, approximate in zip(values, reconstructed):
        if ora.is_value_within_error_bound(eb, float(real), float(approximate)):
            continue
        spacing = float(np.spacing(np.abs(np.float32(real))))
        assert eb.kind == 1 and spacing >= 0.5, (real, approximate)
        assert abs(float(real) - float(approximate)) <= eb.value + spacing, (real, approximate)


def _assert_round_trip(eb, timestamps, values, batch):  # compression.rs:865-929
    ts, reconstructed, rows, _ = ora.grid_batch(batch)
    assert np.array_equal(ts, timestamps)
    _assert_values_within(eb, values, reconstructed)
    assert int(rows.sum()) == len(values)


def test_compress_empty_time_series():  # compression.rs:422-434
    assert len(ora.try_compress_univariate_time_series([], [], LOSSLESS)) == 0


def test_compress_mismatched_lengths():  # compression.rs:202-206
    with pytest.raises(ora.OracleError, match="different lengths"):
        ora.try_compress_univariate_time_series([1, 2], [1.0], LOSSLESS)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("structure,kwargs,eb,expected", [
    ("constant", {}, LOSSLESS, [MDB_PMC_MEAN_ID]),                                   # :436-454
    ("random", {"value_range": (9.8, 10.2)}, ABS_FIVE, [MDB_PMC_MEAN_ID]),            # :456-484
    ("random", {"value_range": (9.8, 10.2)}, REL_FIVE, [MDB_PMC_MEAN_ID]),            # :466-494
    ("linear", {}, LOSSLESS, [MDB_SWING_ID]),                                         # :496-514
    ("linear", {"noise_range": (1.0, 1.05)}, ABS_FIVE, [MDB_SWING_ID]),               # :516-544
    ("linear", {"noise_range": (1.0, 1.05)}, REL_FIVE, [MDB_SWING_ID]),               # :526-554
    ("random", {"value_range": datagen.largest_random_without_overflow()}, LOSSLESS,
     [MDB_MACAQUE_V_ID]),                                                             # :556-574
])
def test_compress_known_segment(irregular, structure, kwargs, eb, expected):  # :576-603
    for seed in range(5):
        rng = np.random.default_rng(100 + seed)
        timestamps = datagen.generate_timestamps(10, irregular, rng)
        values = datagen.generate_values(timestamps, structure, rng, **kwargs)
        batch = ora.try_compress_univariate_time_series(timestamps, values, eb)
        assert batch.model_type_id.tolist() == expected
        _assert_round_trip(eb, timestamps, values, batch)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("generate,expected", [
    ([MDB_MACAQUE_V_ID, MDB_SWING_ID, MDB_PMC_MEAN_ID],
     [MDB_MACAQUE_V_ID, MDB_SWING_ID, MDB_PMC_MEAN_ID]),                              # :605-624
    ([MDB_PMC_MEAN_ID, MDB_SWING_ID, MDB_MACAQUE_V_ID], [MDB_PMC_MEAN_ID, MDB_SWING_ID]),  # :626-645
])
def test_compress_known_time_series(irregular, generate, expected):  # :647-707
    length = 50
    structures = {MDB_PMC_MEAN_ID: "constant", MDB_SWING_ID: "linear", MDB_MACAQUE_V_ID: "random"}
    for seed in range(5):
        rng = np.random.default_rng(200 + seed)
        timestamps = datagen.generate_timestamps(3 * length, irregular, rng)
        parts = []
        for k, model_type_id in enumerate(generate):
            parts.append(datagen.generate_values(
                timestamps[k * length:(k + 1) * length], structures[model_type_id], rng,
                value_range=datagen.largest_random_without_overflow()))
        values = np.concatenate(parts)
        batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
        assert batch.model_type_id.tolist() == expected
        _assert_round_trip(LOSSLESS, timestamps, values, batch)


@pytest.mark.parametrize("irregular", [False, True])
@pytest.mark.parametrize("noise", [None, (1.0, 1.05)])
@pytest.mark.parametrize("eb", [LOSSLESS, ABS_FIVE, REL_FIVE], ids=["lossless", "abs5", "rel5"])
def test_compress_synthetic_time_series(irregular, noise, eb):  # compression.rs:733-863
    timestamps, values = datagen.generate_univariate_time_series(
        50_000, (50, 501), irregular, noise, (100.0, 200.0), seed=300)
    batch = ora.try_compress_univariate_time_series(timestamps, values, eb)
    ts, reconstructed, rows, metrics = ora.grid_batch(batch)
    assert np.array_equal(ts, timestamps)  # bit exact timestamps
    _assert_values_within(eb, values, reconstructed)
    assert metrics["rows_created"] == len(values)
    assert metrics["regular_segments"] + metrics["irregular_segments"] == len(batch)


def test_compress_and_store_residuals_in_a_separate_segment():  # compression.rs:932-978
    timestamps = [100, 200, 300, 400, 500]
    values = [73.0, 37.0, 37.0, 37.0, 73.0]
    batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
    assert len(batch) == 1
    assert batch.model_type_id[0] == MDB_MACAQUE_V_ID
    assert (batch.start_time[0], batch.end_time[0]) == (100, 500)
    assert batch.timestamps.value(0) == bytes([5])
    assert (batch.min_value[0], batch.max_value[0]) == (37.0, 73.0)
    assert len(batch.values.value(0)) == 8
    assert batch.residuals.value(0) == b""
    assert np.isnan(batch.error[0])


def test_residuals_longer_than_255_become_a_separate_segment():  # compression.rs:310-349
    rng = np.random.default_rng(9)
    constant = np.full(20, 5.0, dtype=np.float32)
    noise = rng.uniform(-1e30, 1e30, size=300).astype(np.float32)
    values = np.concatenate([constant, noise])
    timestamps = np.arange(len(values), dtype=np.int64) * 100
    batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
    assert batch.model_type_id.tolist()[:2] == [MDB_PMC_MEAN_ID, MDB_MACAQUE_V_ID]
    assert batch.residuals.value(0) == b""
    _assert_round_trip(LOSSLESS, timestamps, values, batch)
    # 255 residuals still ride in the model's segment.
    values = np.concatenate([constant, noise[:255]])
    timestamps = np.arange(len(values), dtype=np.int64) * 100
    batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
    assert len(batch) == 1 and batch.residuals.value(0)[-1] == 255
    _assert_round_trip(LOSSLESS, timestamps, values, batch)


# ---- operators (crates/modelardb_storage) ---------------------------------------------------------

def test_three_point_series_aggregates():
    # crates/modelardb_embedded/src/operations/data_folder.rs:1087-1109,1165-1234: field_1 =
    # [37,38,39], field_2 = [73,72,71]; COUNT 3, MIN 37/71, MAX 39/73, SUM(field_2) 216? the
    # reference sums field_2 of two series; one series' own sum is 216 and is checked here.
    from modelardb_rs_amd import MDB_AGG_AVG, MDB_AGG_COUNT, MDB_AGG_MAX, MDB_AGG_MIN, MDB_AGG_SUM
    timestamps = [100, 200, 300]
    for values, mn, mx, total in (([37.0, 38.0, 39.0], 37.0, 39.0, 114.0),
                                  ([73.0, 72.0, 71.0], 71.0, 73.0, 216.0)):
        batch = ora.try_compress_univariate_time_series(timestamps, values, LOSSLESS)
        state = ora.agg_batch(batch, MDB_AGG_COUNT | MDB_AGG_MIN | MDB_AGG_MAX | MDB_AGG_SUM)
        assert (state.count, state.min, state.max, state.sum) == (3, mn, mx, total)
        state = ora.agg_batch(batch, MDB_AGG_AVG)
        assert state.sum / state.count == total / 3


def test_segment_aggregates_match_grid_aggregates():
    # crates/modelardb_server/tests/integration_test.rs:1128-1246: COUNT/MIN/MAX exact, SUM/AVG
    # within 0.001 % between segment aggregates and aggregates over the reconstructed points.
    from modelardb_rs_amd import MDB_AGG_COUNT, MDB_AGG_MAX, MDB_AGG_MIN, MDB_AGG_SUM
    mask = MDB_AGG_COUNT | MDB_AGG_MIN | MDB_AGG_MAX | MDB_AGG_SUM
    timestamps, values = datagen.generate_univariate_time_series(
        20_000, (50, 501), False, (1.0, 1.05), (100.0, 200.0), seed=400)
    for eb in (LOSSLESS, REL_FIVE):
        batch = ora.try_compress_univariate_time_series(timestamps, values, eb)
        on_segments = ora.agg_batch(batch, mask)
        on_points = ora.agg_batch_range(batch, -(1 << 62), 1 << 62, mask)
        assert on_segments.count == on_points.count == len(values)
        assert on_segments.min == on_points.min
        assert on_segments.max == on_points.max
        assert abs(on_segments.sum - on_points.sum) <= 1e-5 * abs(on_points.sum)


def test_macaque_v_lossy_subnormal_and_tiny_values():
    # The deviation of a subnormal value underflows to 0, log2(0) is -inf and Rust's saturating
    # `as i32` turns the rewrite position into a huge negative number (macaque_v.rs:185); the oracle
    # clamps it. Around the subnormal boundary the reference's second, unchecked
    # rewrite (macaque_v.rs:190-193) can leave the relative bound (e.g. 1.1754942e-38 -> 1.102e-38
    # under 5 %): that is the reference's behaviour and is restated, not repaired. What must hold:
    # no undefined behaviour, a decodable stream, and normal-range values within the bound.
    values = np.array([1e-45, 2e-45, 1.1754942e-38, 3e-39, -1e-41, 0.0, 5e-40, 1e-30, 1e-45, 37.0],
                      dtype=np.float32)
    for eb in (REL_FIVE, REL_TEN, error_bound("absolute", 1e-30)):
        data, mn, mx, _, _, _ = ora.macaque_v_compress(eb, values)
        decoded = ora.macaque_v_grid(data, len(values))
        assert decoded.min() == mn and decoded.max() == mx   # min/max are of the STORED values
        for real, approximate in zip(values, decoded):
            if abs(real) >= 1e-30:
                assert ora.is_value_within_error_bound(eb, float(real), float(approximate))
