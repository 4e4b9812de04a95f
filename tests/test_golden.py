"""Committed golden fixtures (tests/golden/): the reference's own known-answer vectors, and the
oracle's regression vectors. CPU tests hold the oracle to both; the GPU test holds the HIP path
(fit, grid, aggregates through the C ABI) to the same vectors."""

import json
import os

import numpy as np
import pytest

import oracle_lib as ora
import modelardb_rs_amd as mdb
from modelardb_rs_amd import error_bound

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOSSLESS = error_bound("lossless")
MASK = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _f32(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def _error_bound(name):
    import cases
    return cases.error_bounds()[name]


def test_reference_known_answers_hold_for_the_oracle():
    kats = _load("reference_kats.json")
    assert ora.bits_read(bytes(kats["bits"]["bytes"]), [1] * 24)[0] == kats["bits"]["bits"]
    for case in kats["timestamps_sizes"]:
        data = ora.compress_residual_timestamps(case["timestamps"])
        assert len(data) == case["compressed_len"], case["source"]
        back = ora.decompress_all_timestamps(case["timestamps"][0], case["timestamps"][-1], data)
        assert back.tolist() == case["timestamps"]
    for case in kats["len"]:
        assert ora.seg_len(case["start"], case["end"], bytes(case["timestamps"])) == case["len"]
    for case in kats["macaque_v_state"]:
        _, _, _, lz, tz, _ = ora.macaque_v_compress(LOSSLESS, case["values"])
        assert (lz, tz) == (case["leading"], case["trailing"]), case["source"]
    seg = kats["segment"]
    batch = ora.try_compress_univariate_time_series(seg["timestamps"], seg["values"], LOSSLESS)
    row = batch.rows()[0]
    assert row[:3] == (seg["model_type_id"], seg["start"], seg["end"])
    assert list(row[3]) == seg["timestamps_bytes"] and (row[4], row[5]) == (seg["min"], seg["max"])
    assert len(row[6]) == seg["values_len"] and len(row[7]) == seg["residuals_len"]
    for case in kats["values_column_lengths"]:
        model = ora.fit_next_model(0, LOSSLESS, [100, 200, 300, 400, 500], case["values"])
        assert model.model_type_id == (0 if case["model"] == "pmc_mean" else 1), case["source"]
        assert model.end_index == case["end_index"]
        finished = ora.model_finish(model, LOSSLESS, 4, [100, 200, 300, 400, 500], case["values"])
        assert len(finished.values.value(0)) == case["values_len"], case["source"]
    derived = kats["derived"]
    got = ora.macaque_v_compress(LOSSLESS, derived["macaque_v_lossless"]["values"])[0]
    assert got.hex() == derived["macaque_v_lossless"]["hex"]
    for n, encoded in derived["regular_lengths"].items():
        assert ora.compress_residual_timestamps(list(range(int(n)))).hex() == encoded


def _check_vector(vector, compress, grid, agg, agg_range):
    ts = np.asarray(vector["timestamps"], dtype=np.int64)
    values = _f32(vector["values_bits"])
    batch = compress(ts, values, _error_bound(vector["error_bound"]))
    expected = vector["segments"]
    assert batch.model_type_id.tolist() == expected["model_type_id"], vector["name"]
    assert batch.start_time.tolist() == expected["start_time"]
    assert batch.end_time.tolist() == expected["end_time"]
    assert batch.min_value.view(np.uint32).tolist() == expected["min_value_bits"]
    assert batch.max_value.view(np.uint32).tolist() == expected["max_value_bits"]
    for column in ("timestamps", "values", "residuals"):
        assert [b.hex() for b in getattr(batch, column).to_bytes_list()] == expected[column], (
            vector["name"], column)
    grid_ts, grid_val, rows, metrics = grid(batch)
    assert grid_ts.tolist() == vector["grid_timestamps"], vector["name"]
    assert grid_val.view(np.uint32).tolist() == vector["grid_values_bits"], vector["name"]
    assert rows.tolist() == vector["rows_per_segment"] and metrics == vector["metrics"]
    for state, want in ((agg(batch, MASK), vector["aggregates"]),
                        (agg_range(batch, vector["range"]["lo"], vector["range"]["hi"], MASK),
                         vector["range"])):
        assert state.count == want["count"], vector["name"]
        assert np.float32(state.min).view(np.uint32) == want["min_bits"]
        assert np.float32(state.max).view(np.uint32) == want["max_bits"]
        want_sum = float(want["sum"])
        if np.isfinite(want_sum):
            assert abs(state.sum - want_sum) <= 1e-5 * max(abs(want_sum), 1e-30), vector["name"]
        else:
            assert np.isnan(state.sum) == np.isnan(want_sum)


def test_oracle_reproduces_its_golden_vectors():
    for vector in _load("oracle_vectors.json")["vectors"]:
        _check_vector(vector, ora.try_compress_univariate_time_series, ora.grid_batch, ora.agg_batch,
                      ora.agg_batch_range)


@pytest.mark.gpu
def test_hip_path_reproduces_the_golden_vectors(hip):
    for vector in _load("oracle_vectors.json")["vectors"]:
        _check_vector(vector, hip.try_compress_univariate_time_series, hip.grid_batch, hip.agg_batch,
                      hip.agg_batch_range)


@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_segment(hip):
    seg = _load("reference_kats.json")["segment"]
    row = hip.try_compress_univariate_time_series(seg["timestamps"], seg["values"], LOSSLESS).rows()[0]
    assert row[:3] == (seg["model_type_id"], seg["start"], seg["end"])
    assert list(row[3]) == seg["timestamps_bytes"] and (row[4], row[5]) == (seg["min"], seg["max"])
    assert len(row[6]) == seg["values_len"] and len(row[7]) == seg["residuals_len"]
