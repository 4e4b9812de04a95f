#!/usr/bin/env python3
"""UncompressedDataManager (the ingest side, N4): rows per second through insert_data_points and the one
launch per error bound that compresses the finished buffers. Development tool."""
import os, sys, time
import numpy as np, pyarrow as pa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelardb_rs_amd as mdb  # noqa: E402
from modelardb_rs_amd import host  # noqa: E402
import datagen  # noqa: E402


def main():
    ctx = mdb.Context(0)
    n_series, n_points, rows_per_batch = 64, 1_000_000, 65536 * 16
    values = [datagen.bench_series(s, n_points, 7, 0) for s in range(n_series)]
    bounds = {1: mdb.error_bound("relative", 1.0), 2: mdb.error_bound("relative", 1.0)}
    tags = np.array([f"turbine-{s:04d}" for s in range(n_series)])
    manager = None
    inserted = 0
    seconds_insert = 0.0
    per = rows_per_batch // n_series
    for start in range(0, n_points, per):
        n = min(per, n_points - start)
        # series interleaved row by row, each in time order
        ts = np.tile((np.arange(start, start + n, dtype=np.int64) * 1000)[:, None], (1, n_series)).ravel()
        f1 = np.stack([v[start:start + n] for v in values], axis=1).ravel()
        batch = pa.RecordBatch.from_arrays([
            pa.array(ts, type=pa.int64()).cast(pa.timestamp("us")), pa.array(f1, type=pa.float32()),
            pa.array(f1 * np.float32(2.0), type=pa.float32()),
            pa.array(np.tile(tags, n), type=pa.string_view())], names=["timestamp", "field_1", "field_2", "tag"])
        if manager is None:
            manager = host.UncompressedDataManager(ctx, batch.schema, 0, [1, 2], [3], bounds, buffer_capacity=65536)
        t0 = time.perf_counter()
        manager.insert_data_points(batch)
        seconds_insert += time.perf_counter() - t0
        inserted += batch.num_rows
    manager.flush()
    t0 = time.perf_counter()
    compressed = manager.compress_finished_buffers()
    seconds_compress = time.perf_counter() - t0
    segments = sum(b.num_rows for b in compressed)
    print(f"insert_data_points: {inserted} rows x 2 fields in {seconds_insert * 1e3:.0f} ms = {inserted / seconds_insert / 1e6:.1f} M rows/s; "
          f"compress_finished_buffers: {2 * inserted} values -> {segments} segments in {seconds_compress * 1e3:.0f} ms = "
          f"{2 * inserted / seconds_compress / 1e9:.2f} G values/s")


main()
