"""Segment batches shared by the parity tests: made by the oracle's compressor from seeded series,
covering every model type, residual tails, regular / irregular timestamps and the edge cases the
reference tests (1- and 2-point segments, NaN / infinities, > 255 residuals)."""

import numpy as np

import datagen
import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import error_bound

LOSSLESS = error_bound("lossless")


def error_bounds():
    return {
        "lossless": LOSSLESS,
        "abs5": error_bound("absolute", 5.0),
        "rel5": error_bound("relative", 5.0),
        "rel1": error_bound("relative", 1.0),
        "abs0.01": error_bound("absolute", 0.01),
    }


def synthetic_series(length, irregular, noise, seed, random_value_range=(100.0, 200.0)):
    return datagen.generate_univariate_time_series(length, (50, 501), irregular, noise,
                                                   random_value_range, seed)


def mixed_batch(eb, irregular, seed, length=20_000, noise=(1.0, 1.05)):
    timestamps, values = synthetic_series(length, irregular, noise, seed)
    batch = ora.try_compress_univariate_time_series(timestamps, values, eb)
    return timestamps, values, batch


def edge_case_series():
    """(name, timestamps, values) of small series that stress boundary conditions."""
    nan, inf = float("nan"), float("inf")
    rng = np.random.default_rng(77)
    cases = [
        ("one_point", [1000], [37.0]),
        ("two_points", [1000, 1700], [37.0, 73.0]),
        ("three_points", [100, 200, 300], [37.0, 38.0, 39.0]),
        ("three_irregular", [100, 100, 200], [1.0, 2.0, 3.0]),
        ("constant_8", np.arange(8) * 10, [5.0] * 8),
        ("linear_9", np.arange(9) * 1000 + 1658671178037, np.arange(9) * 42.0 + 42.0),
        ("nan_run", np.arange(12) * 100, [nan] * 12),
        ("inf_run", np.arange(12) * 100, [inf] * 6 + [-inf] * 6),
        ("specials", np.arange(10) * 100, [0.0, -0.0, 1e-45, -1e-45, 1.1754942e-38, nan, inf, -inf,
                                            3.4028235e38, -3.4028235e38]),
        ("kat_73_37", [100, 200, 300, 400, 500], [73.0, 37.0, 37.0, 37.0, 73.0]),
        ("residuals_255", np.arange(275) * 100,
         np.concatenate([np.full(20, 5.0), rng.uniform(-1e30, 1e30, 255)])),
        ("residuals_300", np.arange(320) * 100,
         np.concatenate([np.full(20, 5.0), rng.uniform(-1e30, 1e30, 300)])),
        ("leading_residuals", np.arange(40) * 100,
         np.concatenate([rng.uniform(-1e30, 1e30, 5), np.full(35, 7.0)])),
        ("swing_then_residuals", np.arange(30) * 100,
         np.concatenate([np.arange(20) * 3.0 + 1.0, rng.uniform(-1e3, 1e3, 10)])),
        ("decreasing_swing_residual_min", np.arange(14) * 100,
         np.concatenate([50.0 - np.arange(12) * 2.0, [-3.4028235e38, 3.4028235e38]])),
        ("big_bucket_timestamps", [0, 5, 5 + 70, 5 + 70 + 300, 700 + 2500, 3200 + 3_000_000_000,
                                   3_000_003_300 + (1 << 40), (1 << 41)], np.arange(8) * 1.0),
        ("tiny_values", np.arange(12) * 100, [1e-45, 2e-45, 1.1754942e-38, 3e-39, -1e-41, 0.0, 5e-40,
                                              1e-30, 1e-45, 37.0, -1e-38, 2.5e-38]),
        ("epoch_regular", np.arange(500) * 1000 + 1658671178037,
         100.0 + np.sin(np.arange(500) / 20.0)),
    ]
    return [(name, np.asarray(ts, dtype=np.int64), np.asarray(v, dtype=np.float32))
            for name, ts, v in cases]


def edge_case_batch(eb=LOSSLESS):
    """One batch holding the segments of every edge-case series (as separate chunks)."""
    parts = [ora.try_compress_univariate_time_series(ts, v, eb) for _, ts, v in edge_case_series()]
    return mdb.SegmentBatch.concat(parts)


def assert_grid_equal(got, expected):
    """Timestamps bit-exact; values bit-exact (NaN payloads included)."""
    got_ts, got_val = got[0], got[1]
    exp_ts, exp_val = expected[0], expected[1]
    assert len(got_ts) == len(exp_ts)
    assert np.array_equal(got_ts, exp_ts)
    assert np.array_equal(np.asarray(got_val).view(np.uint32), np.asarray(exp_val).view(np.uint32))
