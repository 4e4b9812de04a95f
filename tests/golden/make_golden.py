#!/usr/bin/env python3
"""Generates tests/golden/*.json.

Two kinds of fixtures:
  reference_kats.json   known-answer vectors that appear as constants in the reference's own unit
                        tests (inputs and expected outputs only, with the reference file:line of the
                        test), plus the two vectors SURVEY 8(c) derives by hand. These are written
                        by hand below - no reference source text is stored.
  oracle_vectors.json   seeded input series and the segments / reconstructed points / aggregates the
                        CPU oracle produces for them. The reference cannot be run in the authoring
                        container (no Rust toolchain), so these are regression vectors of the
                        KAT-pinned oracle, labelled as such. The HIP path must reproduce them exactly.
Run from the repository root: python tests/golden/make_golden.py
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import cases  # noqa: E402
import datagen  # noqa: E402
import oracle_lib as ora  # noqa: E402
import modelardb_rs_amd as mdb  # noqa: E402

F32_MAX = 3.4028234663852886e38

REFERENCE_KATS = {
    "_comment": "paths relative to crates/modelardb_compression/src/ of the reference",
    "bits": {"source": "models/bits.rs:188-208", "bytes": [255, 170, 0],
             "bits": [1] * 8 + [1, 0, 1, 0, 1, 0, 1, 0] + [0] * 8},
    "timestamps_sizes": [
        {"source": "models/timestamps.rs:321-332", "timestamps": [1579701905500 + 100 * i for i in range(5)], "compressed_len": 1},
        {"source": "models/timestamps.rs:335-346", "timestamps": [1579694400057, 1579694400197, 1579694400353, 1579694400493, 1579694400650], "compressed_len": 4},
        {"source": "models/timestamps.rs:349-357", "timestamps": [100, 100, 200], "compressed_len": 1},
        {"source": "models/timestamps.rs:360-369", "timestamps": [100, 37, 38, 200], "compressed_len": 3},
        {"source": "models/timestamps.rs:372-381", "timestamps": [500, 245, 246, 500], "compressed_len": 4},
        {"source": "models/timestamps.rs:384-393", "timestamps": [5000, 2953, 2954, 5000], "compressed_len": 5},
        {"source": "models/timestamps.rs:396-405", "timestamps": [5000000000, 2852516353, 2852516354, 5000000000], "compressed_len": 10},
    ],
    "len": [
        {"source": "models/mod.rs:409-411", "start": 1658671178037, "end": 1658671178037, "timestamps": [], "len": 1},
        {"source": "models/mod.rs:414-416", "start": 1658671178037, "end": 1658671187047, "timestamps": [10], "len": 10},
    ],
    "macaque_v_state": [
        {"source": "models/macaque_v.rs:378-387", "values": [37.0, 73.0], "leading": 8, "trailing": 17},
        {"source": "models/macaque_v.rs:389-398", "values": [37.0, 71.0, 73.0], "leading": 8, "trailing": 17},
    ],
    "segment": {"source": "compression.rs:932-978", "timestamps": [100, 200, 300, 400, 500],
                "values": [73.0, 37.0, 37.0, 37.0, 73.0], "model_type_id": 2, "start": 100, "end": 500,
                "timestamps_bytes": [5], "min": 37.0, "max": 73.0, "values_len": 8, "residuals_len": 0},
    "values_column_lengths": [
        {"source": "types.rs:536-589", "model": "pmc_mean", "values": [10.0, 10.0, 10.0, 10.0, 10.0], "end_index": 4, "values_len": 0},
        {"source": "types.rs:550-561", "model": "pmc_mean", "values": [10.0, 10.0, 10.0, 10.0, -F32_MAX], "end_index": 3, "values_len": 1},
        {"source": "types.rs:564-575", "model": "pmc_mean", "values": [10.0, 10.0, 10.0, 10.0, F32_MAX], "end_index": 3, "values_len": 0},
        {"source": "types.rs:578-589", "model": "pmc_mean", "values": [10.0, 10.0, 10.0, -F32_MAX, F32_MAX], "end_index": 2, "values_len": 4},
        {"source": "types.rs:628-640", "model": "swing", "values": [10.0, 20.0, 30.0, 40.0, 50.0], "end_index": 4, "values_len": 0},
        {"source": "types.rs:643-655", "model": "swing", "values": [10.0, 20.0, 30.0, 40.0, -F32_MAX], "end_index": 3, "values_len": 5},
        {"source": "types.rs:673-685", "model": "swing", "values": [10.0, 20.0, 30.0, -F32_MAX, F32_MAX], "end_index": 2, "values_len": 8},
        {"source": "types.rs:688-700", "model": "swing", "values": [50.0, 40.0, 30.0, 20.0, 10.0], "end_index": 4, "values_len": 1},
        {"source": "types.rs:718-730", "model": "swing", "values": [50.0, 40.0, 30.0, 20.0, F32_MAX], "end_index": 3, "values_len": 5},
        {"source": "types.rs:733-745", "model": "swing", "values": [50.0, 40.0, 30.0, -F32_MAX, F32_MAX], "end_index": 2, "values_len": 8},
    ],
    "derived": {
        "source": "hand-derived in SURVEY 8(c) from macaque_v.rs:76-164 and timestamps.rs:99-108",
        "macaque_v_lossless": {"values": [73.0, 37.0, 37.0, 37.0, 73.0], "hex": "42920000d03c3a43"},
        "regular_lengths": {"3": "03", "5": "05", "127": "7f", "128": "0080", "255": "00ff", "65536": "010000"},
    },
}


def f32_hex(array):
    return np.asarray(array, dtype=np.float32).view(np.uint32).tolist()


def batch_record(batch):
    return {
        "model_type_id": batch.model_type_id.tolist(),
        "start_time": batch.start_time.tolist(),
        "end_time": batch.end_time.tolist(),
        "min_value_bits": f32_hex(batch.min_value),
        "max_value_bits": f32_hex(batch.max_value),
        "timestamps": [b.hex() for b in batch.timestamps.to_bytes_list()],
        "values": [b.hex() for b in batch.values.to_bytes_list()],
        "residuals": [b.hex() for b in batch.residuals.to_bytes_list()],
    }


def oracle_vectors():
    mask = mdb.MDB_AGG_COUNT | mdb.MDB_AGG_MIN | mdb.MDB_AGG_MAX | mdb.MDB_AGG_SUM
    out = []
    inputs = []
    for name, ts, values in cases.edge_case_series():
        inputs.append((name, "lossless", ts, values))
    for eb_name in ("lossless", "abs5", "rel5", "rel1"):
        for irregular in (False, True):
            ts, values = cases.synthetic_series(1500, irregular, (1.0, 1.05), seed=51,
                                                random_value_range=(100.0, 200.0))
            inputs.append((f"synthetic_{eb_name}_{'irregular' if irregular else 'regular'}", eb_name, ts, values))
    ts, values = datagen.sine_series(3, 4000)
    inputs.append(("sine_rel1", "rel1", ts, values))
    for name, eb_name, ts, values in inputs:
        eb = cases.error_bounds()[eb_name]
        batch = ora.try_compress_univariate_time_series(ts, values, eb)
        grid_ts, grid_val, rows, metrics = ora.grid_batch(batch)
        state = ora.agg_batch(batch, mask)
        lo, hi = int(ts[len(ts) // 4]), int(ts[(3 * len(ts)) // 4])
        ranged = ora.agg_batch_range(batch, lo, hi, mask)
        out.append({
            "name": name, "error_bound": eb_name,
            "timestamps": np.asarray(ts).tolist(), "values_bits": f32_hex(values),
            "segments": batch_record(batch),
            "grid_timestamps": grid_ts.tolist(), "grid_values_bits": f32_hex(grid_val),
            "rows_per_segment": rows.tolist(), "metrics": metrics,
            "aggregates": {"count": state.count, "min_bits": f32_hex([state.min])[0],
                           "max_bits": f32_hex([state.max])[0], "sum": repr(state.sum)},
            "range": {"lo": lo, "hi": hi, "count": ranged.count, "min_bits": f32_hex([ranged.min])[0],
                      "max_bits": f32_hex([ranged.max])[0], "sum": repr(ranged.sum)},
        })
    return out


def main():
    with open(os.path.join(HERE, "reference_kats.json"), "w") as f:
        json.dump(REFERENCE_KATS, f, indent=1)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump({"_comment": "generated by tests/golden/make_golden.py from the KAT-pinned CPU oracle",
                   "vectors": oracle_vectors()}, f, separators=(",", ":"))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
